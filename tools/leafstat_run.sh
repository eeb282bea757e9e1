#!/bin/bash
# Development aid (GPU box): how coherent the lanes of a wave are when they reach KD leaves, per workload (a -DFRAY_LEAFSTAT build, tools/build_ab.sh leafstat -DFRAY_LEAFSTAT)
OUT=$1; mkdir -p $OUT
for w in $2; do
  FRAYHIP_LIB=$PWD/build/ab/leafstat/libfrayhip.so timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-serial-pass --workload $w > $OUT/$w.json 2> $OUT/$w.err
  echo "== $w" >> $OUT/leafstat.txt; grep leafstat $OUT/$w.err | tail -4 >> $OUT/leafstat.txt
done
cat $OUT/leafstat.txt
