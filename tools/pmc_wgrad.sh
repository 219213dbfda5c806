#!/bin/bash
# GPU box: counter passes for the grouped weight-gradient kernels -> gpurun_out/pmc_wgrad/{a,b,c} + table.
#   gpurun --timeout 900 -- 'bash tools/pmc_wgrad.sh'
set -e -o pipefail
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_wgrad
rm -rf $OUT; mkdir -p $OUT
cd /tmp
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-families --no-other-configs"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/a -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/a.err
echo "pass a done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/b -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/b.err
echo "pass b done"
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/c.err
echo "pass c done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/d -o p -- python3 $ROOT/bench.py $Q > /dev/null 2> $OUT/d.err
echo "pass d done"
python3 $ROOT/tools/pmc_table.py "g3b_group|g3_reduce|bias_partial" $OUT/a $OUT/b $OUT/c $OUT/d > $OUT/table.txt
cat $OUT/table.txt
