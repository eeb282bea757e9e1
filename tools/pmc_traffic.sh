#!/bin/bash
# HBM traffic of one headline frame: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (they do not
# fit one pass on gfx950), kernel trace only, then summed per kernel.  Run on the GPU box from the repo root.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$c.log 2>&1 || echo "pass $c failed"
done
cd $ROOT
python3 tools/pmc_summarise.py $OUT/pmc_traffic.json $OUT/FETCH_SIZE $OUT/WRITE_SIZE > $OUT/pmc_traffic.txt 2>&1
