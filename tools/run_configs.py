"""GPU-box aid: the other BASELINE configs through bench.py (same timing protocol, same JSON line with the `roofline` object computed from
BASELINE.md section 3's algorithmic figures) — configs[3] 128^3 domain_adaptation, configs[4] 160^3 joint_train in fp16 with dynamic loss scaling
(and in bf16, and with activation recomputation), and configs[1] in the fp32 parity mode.  One child process per configuration (this parent never
touches the GPU); writes one JSON line per configuration.   usage: python tools/run_configs.py [all|da128|joint160|fp32] [out.jsonl] [steps]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "all"
out_path = sys.argv[2] if len(sys.argv) > 2 else None
steps = sys.argv[3] if len(sys.argv) > 3 else "30"
RUNS = [("da128", ["--config", "da128"]),
        ("joint160", ["--config", "joint160"]),
        ("joint160", ["--config", "joint160", "--dtype", "bf16"]),
        ("joint160", ["--config", "joint160", "--dtype", "bf16", "--recompute"]),
        ("fp32", ["--config", "joint96", "--dtype", "fp32"])]
records = []
for tag, extra in RUNS:
    if which not in ("all", tag):
        continue
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "5", "--no-fp32-mode", "--no-cpu-baseline", "--no-other-configs"] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not line:
        print("FAILED: %s\n%s" % (" ".join(extra), r.stderr[-1500:]), flush=True)
        continue
    rec = json.loads(line[-1])
    records.append(rec)
    rf = rec.get("roofline") or {}
    print("%-55s %8.3f ms/step %8.1f volumes/s  %s roofline %.1f %%  peak %.2f GB" % (" ".join(extra), rec["ms_per_step"], rec["value"], rf.get("bound"),
                                                                                   100 * rf.get("frac", 0.0), rec["config"].get("peak_device_memory_GB", 0)), flush=True)
if out_path:
    with open(out_path, "w") as f:
        for r in records:
            f.write(json.dumps(r) + "\n")
