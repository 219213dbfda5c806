"""Per-step kernel list from a rocprofv3 --kernel-trace csv: the launches between the last two optimiser kernels (one replayed step).
usage: python tools/step_trace.py <kernel_trace.csv> [out.json]"""
import csv, json, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sgd_multi_kernel")]
lo, hi = marks[-3], marks[-2]                  # a full step well inside the timed loop
step = rows[lo + 1:hi + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
agg = collections.OrderedDict()
busy = 0
for r in step:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    a = agg.setdefault(r["Kernel_Name"], [0, 0])
    a[0] += 1; a[1] += d
print("one step: %d launches, %.3f ms wall, %.3f ms summed kernel time" % (len(step), (t1 - t0) / 1e6, busy / 1e6))
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4d x %8.2f us = %8.1f us  %s" % (n, d / n / 1e3, d / 1e3, k[:110]))
foreign = [k for k in agg if k.startswith("void at::") or "rocclr" in k]
print("kernels not from libvaeseg inside the step:", foreign if foreign else "none")
if len(sys.argv) > 2:
    json.dump({"launches": len(step), "wall_ms": (t1 - t0) / 1e6, "kernel_ms": busy / 1e6,
               "kernels": {k: {"launches": n, "avg_us": d / n / 1e3} for k, (n, d) in agg.items()}, "foreign": foreign}, open(sys.argv[2], "w"), indent=1)
