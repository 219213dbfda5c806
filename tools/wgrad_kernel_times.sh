#!/bin/bash
# GPU box: per-launch durations of the grouped weight-gradient kernels inside the replayed 96^3 step and the 160^3 step (rocprofv3 --kernel-trace --stats)
#   gpurun --timeout 600 -- 'bash tools/wgrad_kernel_times.sh'
set -e -o pipefail
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/wgrad_times
rm -rf $OUT; mkdir -p $OUT
cd /tmp
Q="--no-cpu-baseline --no-fp32-mode --no-families --no-exchange-forms --no-other-configs"
for cfg in joint96 joint160; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg -o b -- python3 $ROOT/bench.py --config $cfg --steps 20 --warmup 3 $Q > $OUT/$cfg.json 2> $OUT/$cfg.err
  f=$(find $OUT/$cfg -name "*kernel_stats.csv" | head -1)
  echo "== $cfg: $(python3 -c "import json;d=json.load(open('$OUT/$cfg.json'));print(d['ms_per_step'])") ms per step under rocprof"
  grep -E "g3b_group|g3_reduce_group|bias_partial" $f | awk -F, '{printf "%-70s calls %s avg_ns %s\n", substr($1,1,70), $2, $4}'
done
