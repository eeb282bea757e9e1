#!/bin/bash
# Extra counter passes (instruction cache, scalar data cache, memory latencies, FP64 instruction mix) for one bench workload.
# usage: tools/profile_extra.sh TAG [WORKLOAD]    -> gpurun_out/extra_$TAG/pmc_raw.{json,txt}
TAG=${1:-x}
WL=${2:-cornell_pt64}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/extra_$TAG
rm -rf $OUT && mkdir -p $OUT
export FRAYHIP_PT_LANES=1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_TC_REQ" \
           "SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/p$i.log 2>&1 || echo "pmc pass $i failed (see p$i.log)"
  echo "pass $i done" >> $OUT/progress.log
done
cd $ROOT
python3 tools/pmc_summarise.py $OUT/pmc_raw.json $OUT/p* > $OUT/pmc_raw.txt 2>&1
tail -40 $OUT/pmc_raw.txt
