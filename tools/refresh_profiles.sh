#!/bin/bash
# GPU box: regenerate everything profiles/ holds for the current build -> gpurun_out/prof/ (copy the summaries into profiles/ afterwards).
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh'
set -e -o pipefail
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp
Q="--no-cpu-baseline --no-fp32-mode --no-families --no-other-configs"
# 1. the bench line (+ the live-timed launches behind roofline.families)
python3 $ROOT/bench.py --dump-launches $OUT/live_launches.json > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
# 2. per-kernel summary and the launches of one replayed step
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 20 --warmup 3 $Q > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
python3 $ROOT/tools/step_trace.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) $OUT/step_kernels.json > $OUT/step_kernels.txt
echo "stats done"
# 3. HBM traffic (separate passes, MI355X_MICROARCH.md HBM section)
# (--no-families since round 6: the eager family-timing legs put spin_kernel / per-step pack launches into the average — VERDICT r05 weak 8; $Q = graph replay only)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 $Q > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 $Q > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write auto $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt
echo "traffic done"
# 4. matrix-core counters
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 $Q > /dev/null 2> $OUT/pmc_mfma.err
python3 $ROOT/tools/pmc_mfma.py $OUT/pmc_mfma $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/mfma_util.json > $OUT/mfma_util.txt
echo "mfma done"
# 5. the other BASELINE configurations
python3 $ROOT/tools/run_configs.py all $OUT/other_configs.jsonl > $OUT/other_configs.txt 2> $OUT/other_configs.err
head -3 $OUT/step_kernels.txt; head -1 $OUT/hbm_traffic.txt; tail -c 300 $OUT/bench.json
# 6. the fp32 parity mode's per-kernel summary
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats32 -o bench -- python3 $ROOT/bench.py --dtype fp32 --steps 12 --warmup 2 $Q > /dev/null 2> $OUT/stats32.err
cp $(find $OUT/stats32 -name "*kernel_stats.csv" | head -1) $OUT/fp32_mode_kernel_stats.csv
echo "fp32 stats done"
