#!/bin/bash
# Development aid (GPU box): which wave held which 8x8 tile of a k_whitted frame, and when (a -DFRAY_TILESTAT build: tools/build_ab.sh tilestat -DFRAY_TILESTAT).
#   tools/tilestat_run.sh OUTDIR workload [lib]
OUT=$1; W=$2; L=${3:-tilestat}; mkdir -p $OUT
FRAY_TILESTAT_OUT=$OUT/$W.bin FRAYHIP_LIB=$PWD/build/ab/$L/libfrayhip.so timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-serial-pass --workload $W > $OUT/$W.json 2> $OUT/$W.err
python tools/tilestat.py $OUT/$W.bin | tee $OUT/$W.txt
