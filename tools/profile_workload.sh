#!/bin/bash
# Profiles of a bench workload for profiles/ (run on the GPU box from the repo root, e.g. through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `bench.py` with ONE batch lane (FRAYHIP_PT_LANES=1): every launch alone on the chip, so
#      launches x average duration <= frame time and the averages are comparable with bench.py's serialised pass;
#   2. separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ counters in sets of 8; the vector-memory path: TA / TCP; the typed
#      instruction counters: FP64 / FP32 add, mul, fma, transcendental, integer, conversions),
#      kernel trace only, as the pool requires;
#   3. tools/pmc_finish.py folds them into gpurun_out/prof_$TAG/pmc.json (+ the device code's source hash).
# usage: tools/profile_workload.sh TAG [WORKLOAD] [quick]      (quick: the SQ and TA/TCP passes only)
#   WORKLOAD may carry the suffix _contracted: the same workload with `--arith contracted` (option fp_contract = 1); the record is then keyed <workload>_contracted
set -e
TAG=${1:-r03}
WL=${2:-cornell_pt64}
QUICK=${3:-}
KEY=$WL
ARITH=""
case "$WL" in *_contracted) WL=${WL%_contracted}; ARITH="--arith contracted";; esac
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT      # a re-run must not mix with an older run's files
export FRAYHIP_PT_LANES=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-pass --no-contracted $ARITH --workload $WL > $OUT/trace_bench.json 2> $OUT/trace.log || echo "trace pass failed"
echo "trace done" >> $OUT/progress.log
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
      "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
      "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TOTAL_WRITE_sum"
      # the vector memory path: how busy the texture-address unit is, what a read / a write instruction costs the L1 in accesses (read and write
      # counted apart and calibrated by tools/ubench/l1_access.hip), and where the path waits (TCP_GATE_EN1, round 3's "tcp_busy", is a clock-gate
      # enable: it read 0.975 for a kernel without a single vector load and is gone)
      "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_sum"
      "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
      # the measured instruction mix (what tools/pmc_mix.sh collects on its own)
      "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH"
      "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT")
if [ -z "$QUICK" ]; then SETS+=("FETCH_SIZE" "WRITE_SIZE"); fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-contracted $ARITH --workload $WL > $OUT/p$i.log 2>&1 || echo "pmc pass $i failed (see p$i.log)"
  echo "pass $i done" >> $OUT/progress.log
done
cd $ROOT
python3 tools/pmc_summarise.py $OUT/pmc_raw.json $OUT/p* > $OUT/pmc_raw.txt 2>&1
python3 tools/pmc_finish.py $OUT $KEY > $OUT/pmc.txt 2>&1
cat $OUT/pmc.txt
# keep what profiles/ wants in a few small files (the per-pass directories stay on the box's scratch copy)
cp "$(ls -t $OUT/trace/*/*_kernel_stats.csv | head -1)" $OUT/kernel_stats.csv 2>/dev/null || true
cp "$(ls -t $OUT/trace/*/*_kernel_trace.csv | head -1)" $OUT/kernel_trace.csv 2>/dev/null || true
mkdir -p $OUT/logs && cp $OUT/p[0-9]*.log $OUT/logs/ 2>/dev/null || true
rm -rf $OUT/p[0-9]* $OUT/trace
