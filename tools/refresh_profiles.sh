#!/bin/bash
# GPU box: regenerate what profiles/ holds for the current build -> gpurun_out/prof/ (copy the summaries into profiles/ afterwards).
#   gpurun --timeout 900 -- 'bash tools/refresh_profiles.sh'
set -e -o pipefail
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-mode > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-mode > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write auto $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt
find $OUT -name "*kernel_stats.csv" -o -name "*domain_stats.csv" | head
tail -c 400 $OUT/bench.json
