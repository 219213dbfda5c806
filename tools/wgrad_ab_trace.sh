#!/bin/bash
# GPU box: durations and HBM traffic of the weight-gradient launches of the replayed 96^3 step under two environments (same box, back to back)
#   gpurun --timeout 900 -- 'bash tools/wgrad_ab_trace.sh "VS_WGRAD_BIG=0" "VS_WGRAD_BIG=1"'
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/wgrad_ab
rm -rf $OUT; mkdir -p $OUT
cd /tmp
Q="--no-cpu-baseline --no-fp32-mode --no-families --no-exchange-forms --no-other-configs"
CFG=${CFG:-joint96}
i=0
for v in "$@"; do
  i=$((i+1))
  ( export $v
    rocprofv3 --kernel-trace --output-format csv -d $OUT/t$i -o b -- python3 $ROOT/bench.py --config $CFG --steps 10 --warmup 3 $Q > $OUT/t$i.json 2> $OUT/t$i.err
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f$i -o b -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 $Q > /dev/null 2> $OUT/f$i.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w$i -o b -- python3 $ROOT/bench.py --config $CFG --steps 3 --warmup 1 $Q > /dev/null 2> $OUT/w$i.err )
  echo "== $v"
  python3 $ROOT/tools/wgrad_ab_report.py $OUT/t$i $OUT/f$i $OUT/w$i
done
