#!/bin/bash
# usage: stamps_run.sh OUTDIR "libs" workload
OUT=$1; mkdir -p $OUT
for l in $2; do
  FRAYHIP_PT_LANES=1 FRAYHIP_LIB=$PWD/build/ab/$l/libfrayhip.so timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-serial-pass --workload $3 > $OUT/$l.json 2> $OUT/$l.err
  echo "== $l" >> $OUT/stamps.txt; grep "stamps" $OUT/$l.err | tail -30 >> $OUT/stamps.txt
done
cat $OUT/stamps.txt
