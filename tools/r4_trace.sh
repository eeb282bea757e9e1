#!/bin/bash
# Development aid (GPU box): rocprofv3 kernel statistics of one workload on round 4's tree (build/r4tree) and on this tree, one batch lane.
W=$1; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4trace; mkdir -p $OUT
export FRAYHIP_PT_LANES=1
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/t4 $OUT/t5
(cd $ROOT/build/r4tree && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t4 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-pass --workload $W > $OUT/r4_$W.json 2> $OUT/r4_$W.err)
(cd $ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t5 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-pass --no-contracted --workload $W > $OUT/r5_$W.json 2> $OUT/r5_$W.err)
cp "$(ls -t $OUT/t4/*/*_kernel_stats.csv | head -1)" $OUT/r4_${W}_kernel_stats.csv
cp "$(ls -t $OUT/t5/*/*_kernel_stats.csv | head -1)" $OUT/r5_${W}_kernel_stats.csv
rm -rf $OUT/t4 $OUT/t5
