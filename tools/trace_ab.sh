#!/bin/bash
# Development aid (GPU box, repo root): rocprofv3 kernel statistics of a bench workload for several builds of the library (tools/build_ab.sh), one batch lane.
#   tools/trace_ab.sh OUTDIR "lib1 lib2" WORKLOAD [extra bench args]
OUT=$1; LIBS=$2; WL=$3; shift 3
ROOT=$(pwd); mkdir -p $OUT; OUT=$(cd $OUT && pwd)
export FRAYHIP_PT_LANES=1
cd /tmp && export TMPDIR=/tmp
for l in $LIBS; do
  if [ "$l" = "default" ]; then unset FRAYHIP_LIB; else export FRAYHIP_LIB=$ROOT/build/ab/$l/libfrayhip.so; fi
  rm -rf $OUT/t_$l
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$l -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-pass --workload $WL "$@" > $OUT/$l.json 2> $OUT/$l.err || { echo "$l failed"; exit 1; }
  cp "$(ls -t $OUT/t_$l/*/*_kernel_stats.csv | head -1)" $OUT/${l}_${WL}_kernel_stats.csv
  rm -rf $OUT/t_$l
  echo "== $l"; head -8 $OUT/${l}_${WL}_kernel_stats.csv | cut -d, -f1-4
done
