"""Print the kernels of ONE replayed step of a rocprofv3 --kernel-trace CSV in launch order: start offset, duration, gap to
the previous kernel's end.  usage: trace_timeline.py trace.csv [which_step_from_end=2] [marker=sgd_multi_kernel]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marker = sys.argv[3] if len(sys.argv) > 3 else "sgd_multi_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = marks[-back - 1] + 1, marks[-back] + 1
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = t0
busy = 0.0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    print("%9.1f us  dur %7.1f  gap %6.1f  grid %7s x %-4s  %s" % ((s - t0) * 1e-3, (e - s) * 1e-3, (s - prev_end) * 1e-3,
                                                              r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), name[:80]))
    busy += (e - s) * 1e-3
    prev_end = max(prev_end, e)
print("step span %.1f us, sum of durations %.1f us, %d launches" % ((prev_end - t0) * 1e-3, busy, hi - lo))
