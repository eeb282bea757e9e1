#!/bin/bash
# Profiles of the bench workload for profiles/ (run on the GPU box from the repo root, e.g. through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `bench.py` with ONE batch lane (FRAYHIP_PT_LANES=1): every launch alone on the chip, so
#      launches x average duration <= frame time and the averages are comparable with bench.py's serialised pass;
#   2. separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ counters in sets of 8), kernel trace only, as the pool requires;
#   3. tools/pmc_finish.py folds them into gpurun_out/prof_$TAG/pmc.json (+ the device code's source hash).
# usage: tools/profile_headline.sh TAG [WORKLOAD]
set -e
TAG=${1:-r02}
WL=${2:-cornell_pt64}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT      # a re-run must not mix with an older run's files
export FRAYHIP_PT_LANES=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/trace_bench.json 2> $OUT/trace.log || echo "trace pass failed"
echo "trace done" >> $OUT/progress.log
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $OUT/p$i.log 2>&1 || echo "pmc pass $i failed (see p$i.log)"
  echo "pass $i done" >> $OUT/progress.log
done
cd $ROOT
python3 tools/pmc_summarise.py $OUT/pmc_raw.json $OUT/p* > $OUT/pmc_raw.txt 2>&1
python3 tools/pmc_finish.py $OUT $WL > $OUT/pmc.txt 2>&1
cat $OUT/pmc.txt
