"""Per-kernel HBM traffic from two rocprofv3 passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE), as MI355X_MICROARCH.md §HBM
prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so it is doubled.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json]"""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    rows = list(csv.DictReader(open((glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0])))
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    f, n = fetch[k]
    w = write.get(k, (0.0, 0))[0]
    out[k] = {"launches_sampled": n, "fetch_kib_raw": f, "write_kib": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:12]:
    print("%-70s n=%5d  %10.2f MB/launch (fetch x2 %.2f MB, write %.2f MB)" % (k[:70], v["launches_sampled"], v["hbm_bytes_per_launch"] / 1e6,
                                                                             2 * v["fetch_kib_raw"] * 1024 / 1e6, v["write_kib"] * 1024 / 1e6))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
