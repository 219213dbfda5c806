"""Per-kernel matrix-core utilisation from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16
[+ GRBM_GUI_ACTIVE]) and the kernel durations of an UNPERTURBED --kernel-trace --stats run of the same command:
  executed MFMA FLOP per launch = MOPS x 512 / launches            (rocprofv3's own MfmaFlops* definition)
  MFMA utilisation              = executed FLOP / (average duration x 2.5 PFLOP/s dense bf16 peak)
  busy share                    = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of SIMD cycles, on CUs that hold a wave, in which the
                                  matrix pipe is busy
usage: python tools/pmc_mfma.py <pmc_dir> <kernel_stats.csv> [out.json]"""
import collections, csv, glob, json, sys

PEAK = 2.5e15
rows = list(csv.DictReader(open((glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
        cnt[r["Kernel_Name"]] += 1
dur = {r["Name"]: float(r["AverageNs"]) * 1e-9 for r in csv.DictReader(open(sys.argv[2]))}
out = {}
for k, c in acc.items():
    n = cnt[k]
    mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)
    if not n or not mops:
        continue
    flop = mops * 512.0 / n
    d = dur.get(k)
    out[k] = {"launches_sampled": n, "executed_mfma_flop_per_launch": flop, "avg_duration_us": None if d is None else d * 1e6,
              "mfma_util_of_2.5PF": None if d is None else flop / d / PEAK,
              "mfma_busy_share_of_simd_cycles_on_busy_cus": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * c["SQ_BUSY_CU_CYCLES"]) if c.get("SQ_BUSY_CU_CYCLES") else None}
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["executed_mfma_flop_per_launch"] * kv[1]["launches_sampled"])):
    print("%-74s n=%4d  %7.2f GFLOP/launch  %7.2f us  util %s  busy share %s" % (
        k[:74], v["launches_sampled"], v["executed_mfma_flop_per_launch"] / 1e9, v["avg_duration_us"] or 0.0,
        "%.4f" % v["mfma_util_of_2.5PF"] if v["mfma_util_of_2.5PF"] is not None else "n/a",
        "%.3f" % v["mfma_busy_share_of_simd_cycles_on_busy_cus"] if v["mfma_busy_share_of_simd_cycles_on_busy_cus"] is not None else "n/a"))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
