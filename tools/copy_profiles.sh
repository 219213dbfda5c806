#!/bin/bash
# container: copy the summaries tools/refresh_profiles.sh left under gpurun_out/prof/ into profiles/ under this round's names
#   bash tools/copy_profiles.sh r06
set -e
R=${1:?round prefix, e.g. r06}
S=gpurun_out/prof
cp $S/bench.json profiles/${R}_bench.json
cp $S/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $(find $S/stats -name "*kernel_stats.csv" | head -1) profiles/${R}_bench_bf16_kernel_stats.csv
cp $(find $S/stats -name "*domain_stats.csv" | head -1) profiles/${R}_bench_bf16_domain_stats.csv
cp $S/fp32_mode_kernel_stats.csv profiles/${R}_fp32_mode_kernel_stats.csv
for f in hbm_traffic.json hbm_traffic.txt live_launches.json mfma_util.json mfma_util.txt other_configs.jsonl step_kernels.json step_kernels.txt; do cp $S/$f profiles/${R}_$f; done
