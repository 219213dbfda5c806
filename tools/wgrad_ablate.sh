#!/bin/bash
# GPU box: what the tile loop of the grouped weight-gradient kernels is made of — the 96^3 step under rocprofv3 with parts of g3b_body switched off
# (diagnostic build: VS_STAMPS_DEF=VS_G3B_ABLATE VS_STAMPS_OUT=libvaeseg_ablate.so bash tools/build_stamps.sh; cp tools/_dbg/libvaeseg_ablate.so tools/_stamps/)
set -e -o pipefail
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/wgrad_ablate
rm -rf $OUT; mkdir -p $OUT
cd /tmp
export VS_LIBVAESEG=$ROOT/tools/_stamps/libvaeseg_ablate.so
Q="--no-cpu-baseline --no-fp32-mode --no-families --no-exchange-forms --no-other-configs"
for flag in ${FLAGS:-0 1 4 8 16 32 5 9 13 29 61}; do
  export VS_G3B_ABLATE=$flag
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f$flag -o b -- python3 $ROOT/bench.py --steps 10 --warmup 2 $Q > $OUT/f$flag.json 2> $OUT/f$flag.err
  f=$(find $OUT/f$flag -name "*kernel_stats.csv" | head -1)
  echo "== ablate=$flag"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "g3b_group" in r["Name"]:
        print("   %-62s avg %7.1f us" % (r["Name"][:62], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/f$flag
done
