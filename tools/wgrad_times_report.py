#!/usr/bin/env python3
"""Fold gpurun_out/wgrad_times/<cfg>/b_kernel_stats.csv (tools/wgrad_kernel_times.sh) into the weight-gradient rows: name, calls, average us."""
import csv, json, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wgrad_times"
for cfg in ("joint96", "joint160"):
    f = os.path.join(root, cfg, "b_kernel_stats.csv")
    if not os.path.exists(f):
        continue
    rows = list(csv.DictReader(open(f)))
    print(cfg, "%.3f ms per step under rocprofv3" % json.load(open(os.path.join(root, cfg + ".json")))["ms_per_step"])
    tot = 0.0
    for r in rows:
        if any(k in r["Name"] for k in ("g3b_group", "g3b_uber", "g3_reduce_group", "bias_partial")):
            print("  %-66s %3s calls  avg %7.1f us" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3))
            tot += float(r["AverageNs"]) / 1e3
    print("  weight-gradient launches, total %.1f us per step" % tot)
