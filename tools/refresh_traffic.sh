#!/bin/bash
# GPU box: re-take ONLY the HBM-traffic summary (two PMC passes, graph replay only) and a default bench line behind it -> gpurun_out/prof2/hbm_traffic.{json,txt},
# gpurun_out/bench_after_traffic.json.  For a change that touches the kernel sources after tools/refresh_profiles.sh ran: the bench line's `traffic` carries the hash of
# the sources it was measured on (profiling.sources_sha).  Copy the two files to profiles/rNN_hbm_traffic.* afterwards.
#   gpurun --timeout 900 -- 'bash tools/refresh_traffic.sh'
set -e -o pipefail
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
Q="--no-cpu-baseline --no-fp32-mode --no-families --no-other-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 $Q > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 $Q > /dev/null 2> $OUT/pmc_write.err
python3 $ROOT/tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write auto $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt
head -2 $OUT/hbm_traffic.txt
cp $OUT/hbm_traffic.json $ROOT/profiles/r06_hbm_traffic.json
cd $ROOT && python3 bench.py > $ROOT/gpurun_out/bench_after_traffic.json 2> $ROOT/gpurun_out/bench_after_traffic.err
python3 -c "
import json;d=json.load(open('$ROOT/gpurun_out/bench_after_traffic.json'));print(d['value'],d['ms_per_step'],d['roofline']['traffic'],d['roofline']['traffic_stale'])"
rm -rf $OUT/pmc_fetch $OUT/pmc_write
