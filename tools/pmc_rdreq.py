"""Read-request size classes per kernel (gfx950: TCC_EA0_RDREQ_32B / _64B / _128B, counter_defs.yaml events 28 / 44 / 45) — the calibration
MI355X_MICROARCH.md's HBM section asks for before an absolute FETCH_SIZE figure is trusted on an access pattern other than a wide coalesced stream.
FETCH_SIZE's gfx950 expression tallies a 128-byte request at 64 bytes unless TCC_BUBBLE sees it (hence the guide's "double it" for streams, which are
all 128-byte requests); a kernel whose loads leave 64-byte requests (partial lines: halo rows, 16-byte fragments with gaps) is OVER-counted by the
doubling.  exact = 32 n32 + 64 n64 + 128 n128.
usage: python tools/pmc_rdreq.py <rdreq_dir> <fetch_dir> <steps | auto> [out.json]   (fetch_dir: the --pmc FETCH_SIZE pass of the same command)"""
import collections, csv, glob, json, sys


def per_kernel(d, counters):
    rows = list(csv.DictReader(open((glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0])))
    acc = {c: collections.defaultdict(list) for c in counters}
    for r in rows:
        if r["Counter_Name"] in acc:
            acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {c: {k: (sum(v) / len(v), len(v)) for k, v in kv.items()} for c, kv in acc.items()}


names = ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_sum"]
rd = per_kernel(sys.argv[1], names)
fetch = per_kernel(sys.argv[2], ["FETCH_SIZE"])["FETCH_SIZE"]
kernels = list(rd[names[3]])
if sys.argv[3] == "auto":
    steps = float(max(n for k, (f, n) in rd[names[3]].items() if k.startswith("g3_reduce_group_kernel")))
else:
    steps = float(sys.argv[3])
out, tot_exact, tot_x2 = {}, 0.0, 0.0
for k in kernels:
    n32, n64, n128, nall = (rd[c].get(k, (0.0, 0))[0] for c in names)
    n = rd[names[3]][k][1]
    exact = 32 * n32 + 64 * n64 + 128 * n128
    f2 = 2 * 1024 * fetch.get(k, (0.0, 0))[0]
    out[k] = {"launches_per_step": n / steps, "n32": n32, "n64": n64, "n128": n128, "rdreq": nall, "fetch_bytes_by_class": exact, "fetch_size_x2_bytes": f2}
    if "spin_kernel" not in k:
        tot_exact += exact * n / steps
        tot_x2 += f2 * n / steps
out["_step_fetch_bytes_by_class"], out["_step_fetch_size_x2_bytes"] = tot_exact, tot_x2
print("fetched bytes per step: by request class %.3f GB; FETCH_SIZE x 2 %.3f GB" % (tot_exact / 1e9, tot_x2 / 1e9))
print("%-64s %7s %10s %10s  %5s %5s %5s" % ("kernel", "/step", "class MB", "x2 MB", "%32B", "%64B", "%128B"))
for k, v in sorted(((k, v) for k, v in out.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["fetch_bytes_by_class"] * kv[1]["launches_per_step"])[:20]:
    e = max(v["fetch_bytes_by_class"], 1.0)
    print("%-64s %7.1f %10.2f %10.2f  %5.1f %5.1f %5.1f" % (k[:64], v["launches_per_step"], v["fetch_bytes_by_class"] / 1e6, v["fetch_size_x2_bytes"] / 1e6,
                                                           100 * 32 * v["n32"] / e, 100 * 64 * v["n64"] / e, 100 * 128 * v["n128"] / e))
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
