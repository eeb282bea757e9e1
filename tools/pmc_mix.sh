#!/bin/bash
# Measured instruction mix of a bench workload (run on the GPU box from the repo root): which vector instructions the kernels issue --
# FP64 add / mul / fma / transcendental, the same for FP32, 32- and 64-bit integer, conversions -- beside the totals (VALU, SALU, SMEM, waves).
# Separate --pmc passes with the kernel trace only, as the pool requires; one serialised frame per pass (FRAYHIP_PT_LANES=1).
#   tools/pmc_mix.sh OUTDIR [WORKLOAD] [LIB]        LIB: a name under build/ab/ (tools/build_ab.sh) or "default"
set -e
OUT=$1; WL=${2:-cornell_pt64}; LIB=${3:-default}
ROOT=$(pwd)
mkdir -p $OUT
OUT=$(cd $OUT && pwd)
if [ "$LIB" != "default" ]; then export FRAYHIP_LIB=$ROOT/build/ab/$LIB/libfrayhip.so; fi
export FRAYHIP_PT_LANES=1
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES"
      "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU"
      "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rm -rf $OUT/m$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/m$i -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serial-pass --workload $WL > $OUT/m$i.log 2>&1 || echo "pmc pass $i failed (see m$i.log)"
  echo "mix pass $i done" >> $OUT/progress.log
done
cd $ROOT
python3 tools/pmc_summarise.py $OUT/mix_raw_${LIB}_$WL.json $OUT/m[0-9]* > $OUT/mix_raw_${LIB}_$WL.txt 2>&1
python3 tools/pmc_mix.py $OUT/mix_raw_${LIB}_$WL.json > $OUT/mix_${LIB}_$WL.txt
cat $OUT/mix_${LIB}_$WL.txt
rm -rf $OUT/m[0-9]*
