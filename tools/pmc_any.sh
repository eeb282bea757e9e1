#!/bin/bash
# Development aid (GPU box, repo root): arbitrary counter sets on a bench workload, one rocprofv3 --pmc pass per set (kernel trace only, as the pool requires).
#   tools/pmc_any.sh OUTDIR WORKLOAD LANES "SET 1 COUNTERS" ["SET 2 COUNTERS" ...]      LANES: 1 = serialised launches, 4 = as shipped
set -e
OUT=$1; WL=$2; LANES=$3; shift 3
ROOT=$(pwd); mkdir -p $OUT; OUT=$(cd $OUT && pwd)
export FRAYHIP_PT_LANES=$LANES
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf $OUT/a$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/a$i -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-serial-pass --workload $WL > $OUT/a$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/a$i.log)"
done
cd $ROOT
python3 tools/pmc_summarise.py $OUT/any_${WL}_lanes$LANES.json $OUT/a[0-9]* > $OUT/any_${WL}_lanes$LANES.txt 2>&1
grep -A40 "k_pt_bounce<0, false, false>\|k_pt_shadow<0>" $OUT/any_${WL}_lanes$LANES.txt | grep -v "^void k_pt_bounce<1\|<0, false, true>" | head -60
rm -rf $OUT/a[0-9]*
