"""Per-kernel HBM traffic from two rocprofv3 passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE), as MI355X_MICROARCH.md §HBM
prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so it is doubled.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <step_equivalents_in_the_run | auto> [out.json]
(step equivalents: every forward+backward pass of the bf16 step in the run — bench.py's eager warm-up and family-timing passes, --warmup,
--steps; launches per step = sampled launches / that.  auto: the launches of the once-per-pass slab reduction of the grouped weight gradients)"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(d, counter):
    rows = list(csv.DictReader(open((glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0])))
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
if sys.argv[3] == "auto":
    steps = float(max(n for k, (f, n) in fetch.items() if k.startswith("g3_reduce_group_kernel")))          # one launch per pass
else:
    steps = float(sys.argv[3])
out, total = {}, 0.0
for k in fetch:
    f, n = fetch[k]
    w = write.get(k, (0.0, 0))[0]
    b = (2.0 * f + w) * 1024.0
    out[k] = {"launches_sampled": n, "launches_per_step": n / steps, "fetch_kib_raw": f, "write_kib": w, "hbm_bytes_per_launch": b}
    if "spin_kernel" not in k:
        total += b * n / steps
out["_step_total_bytes"] = total
try:                                   # which kernel sources this was measured on (bench.py flags `traffic_stale` when they differ from the tree's)
    from vae_segmentation_amd import profiling
    out["_sources_sha16"] = profiling.sources_sha()
except Exception as exc:               # the summary is still worth having
    print("no source hash:", exc)
print("HBM traffic per step (FETCH_SIZE x2 + WRITE_SIZE over all kernels): %.3f GB" % (total / 1e9))
for k, v in sorted(((k, v) for k, v in out.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:14]:
    print("%-72s %6.1f/step %9.2f MB/launch (fetch x2 %.2f MB, write %.2f MB)" % (k[:72], v["launches_per_step"], v["hbm_bytes_per_launch"] / 1e6,
                                                                                  2 * v["fetch_kib_raw"] * 1024 / 1e6, v["write_kib"] * 1024 / 1e6))
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
