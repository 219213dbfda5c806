#!/usr/bin/env python3
"""Same-box A/B of bench.py under different environment settings, alternating rounds (box-to-box spread is ~2 %, so only same-box
alternating numbers decide).  usage: python tools/ab.py [--rounds 3] [--steps 200] [--args "--dtype bf16"] "A=1 B=2" "A=0" ...
Each variant is a space-separated list of NAME=VALUE (or "-" for the plain environment).  The parent never touches the GPU."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--args", default="")
ap.add_argument("--out", default=None)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
res = {v: [] for v in a.variants}
for r in range(a.rounds):
    for v in a.variants:
        env = dict(os.environ)
        if v != "-":
            for kv in v.split():
                k, val = kv.split("=", 1)
                env[k] = val
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--warmup", "10", "--no-cpu-baseline", "--no-fp32-mode",
               "--no-families", "--no-other-configs"] + a.args.split()
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("variant %r failed:\n%s" % (v, out.stderr[-2000:]), flush=True)
            res[v].append(float("nan"))
            continue
        rec = json.loads(line[-1])
        res[v].append(rec["ms_per_step"])
        print("round %d  %-60s %.4f ms  loss %.6f" % (r, v, rec["ms_per_step"], rec["config"].get("final_loss", float("nan"))), flush=True)
print()
for v in a.variants:
    xs = sorted(res[v])
    print("%-60s median %.4f  all %s" % (v, xs[len(xs) // 2], " ".join("%.4f" % x for x in res[v])))
if a.out:
    with open(a.out, "w") as f:
        json.dump({"steps": a.steps, "rounds": a.rounds, "args": a.args, "ms_per_step": res}, f, indent=1)
