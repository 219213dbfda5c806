"""Per-kernel matrix-core utilisation from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES (+ any other SQ counters
collected in the same pass): summed over the launches of each kernel, ratio = MFMA busy cycles / SQ busy cycles.
usage: python tools/pmc_mfma.py <pmc_dir> [out.json]"""
import collections, csv, glob, json, sys

rows = list(csv.DictReader(open((glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        cnt[r["Kernel_Name"]] += 1
out = {}
for k, c in acc.items():
    busy = c.get("SQ_BUSY_CYCLES", 0.0)
    out[k] = {"launches_sampled": cnt[k], **{n: v for n, v in c.items()},
              "mfma_busy_over_sq_busy": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / busy) if busy else None}
for k, v in sorted(out.items(), key=lambda kv: -(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)))[:24]:
    r = v["mfma_busy_over_sq_busy"]
    print("%-80s n=%4d  MFMA busy / SQ busy = %s" % (k[:80], v["launches_sampled"], "%.4f" % r if r is not None else "n/a"))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
