"""Per-kernel table of raw PMC counters from one or more rocprofv3 --pmc passes: mean per launch of every counter, for the kernels whose name contains
one of the patterns.  usage: python tools/pmc_table.py "pattern1|pattern2" <pmc_dir> [<pmc_dir> ...]"""
import collections, csv, glob, sys

pats = sys.argv[1].split("|")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for d in sys.argv[2:]:
    files = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    if not files:
        print("no counter_collection.csv under", d)
        continue
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"]
        if not any(p in k for p in pats):
            continue
        k = k.replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print("== %s" % k)
    for c in sorted(acc[k]):
        n = cnt[k][c]
        print("   %-40s %16.1f  per launch (%d launches)" % (c, acc[k][c] / n, n))
