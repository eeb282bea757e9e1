#!/bin/bash
# SQ counter passes over two headline frames (run on the GPU box from the repo root).
# Each pass is its own rocprofv3 run with --kernel-trace only, as the pool requires.
set -e
ROOT=$(pwd)
WL=${1:-cornell_pt64}
OUT=$ROOT/gpurun_out/pmc_sq_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/p$i.log 2>&1 || echo "pass $i failed (see p$i.log)"
  echo "pass $i done" >> $OUT/progress.log
done
cd $ROOT
python3 tools/pmc_summarise.py $OUT/sq_summary.json $OUT/p* > $OUT/sq_summary.txt 2>&1
