"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import collections, csv, glob, sys
path = (glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0]
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    meta[k] = (r["Grid_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
for k, v in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    waves = int(meta[k][0]) / 64
    print(k, "grid", meta[k][0], "vgpr", meta[k][1], "agpr", meta[k][2], "lds", meta[k][3])
    for c, x in sorted(v.items()):
        m = sum(x) / len(x)
        print("   %-28s %14.0f   per wave %10.1f" % (c, m, m / waves))
