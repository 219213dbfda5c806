"""Fold a rocprofv3 --kernel-trace CSV into per-(kernel, grid) time per step. usage: trace_breakdown.py trace.csv steps_equiv [top]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    a = agg[key]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
tot = sum(a[1] for a in agg.values())
print("total kernel time per step: %.3f ms over %.1f launches" % (tot / steps * 1e-3, sum(a[0] for a in agg.values()) / steps))
byk = collections.defaultdict(lambda: [0, 0.0])
for (name, gx, gy), a in agg.items():
    byk[name][0] += a[0]; byk[name][1] += a[1]
print("---- per kernel")
for name, a in sorted(byk.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-70s %6.1f launches/step  avg %7.1f us  %7.3f ms/step" % (name[:70], a[0] / steps, a[1] / a[0], a[1] / steps * 1e-3))
print("---- per (kernel, grid)")
for (name, gx, gy), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-58s grid %8s x %3s  %5.1f/step  avg %7.1f us  %7.3f ms/step" % (name[:58], gx, gy, a[0] / steps, a[1] / a[0], a[1] / steps * 1e-3))
