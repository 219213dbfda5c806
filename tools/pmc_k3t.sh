#!/bin/bash
# GPU box: counter passes for the 8-channel full-resolution conv kernels (k3t_kernel) -> gpurun_out/pmc_k3t/{a,b,c,d} + table.
#   gpurun --timeout 900 -- 'bash tools/pmc_k3t.sh'            (CFG=joint160 for the 160^3 step)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFG=${CFG:-joint96}
OUT=$ROOT/gpurun_out/pmc_k3t_$CFG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
Q="--config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-families --no-other-configs"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/a -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/b -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/c.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/d -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/d.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/e -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/e.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/f -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/f.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -o p -- python3 $ROOT/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode --no-families --no-other-configs > /dev/null 2> $OUT/s.err
python3 $ROOT/tools/pmc_table.py "k3t_kernel" $OUT/a $OUT/b $OUT/c $OUT/d $OUT/e $OUT/f > $OUT/table.txt
grep k3t_kernel $(find $OUT/s -name "*kernel_stats.csv" | head -1) | awk -F, '{printf "%-80s calls %s avg_ns %s\n", substr($1,1,80), $2, $4}' >> $OUT/table.txt
cat $OUT/table.txt
