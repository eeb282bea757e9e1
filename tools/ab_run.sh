#!/bin/bash
# Development aid (run on the GPU box from the repo root): times workloads on several builds of the library (tools/build_ab.sh).
#   tools/ab_run.sh OUTDIR "lib1 lib2 ..." "workload1 workload2 ..." [steps]
# A lib is a name under build/ab/ or "default".  One line per (lib, workload) in OUTDIR/summary.txt.
OUT=$1; LIBS=$2; WLS=$3; STEPS=${4:-10}
mkdir -p $OUT
for w in $WLS; do
  for l in $LIBS; do
    if [ "$l" = "default" ]; then unset FRAYHIP_LIB; else export FRAYHIP_LIB=$PWD/build/ab/$l/libfrayhip.so; fi
    timeout -k 10 300 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --workload $w > $OUT/${l}_$w.json 2> $OUT/${l}_$w.err || { echo "$l $w FAILED" >> $OUT/summary.txt; continue; }
    python - "$OUT/${l}_$w.json" "$l" "$w" >> $OUT/summary.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]; s = d.get("roofline_shadow_kernel") or {}
print("%-10s %-18s %9.3f ms  %9.1f Mrays/s  %s %.3f (%.3f ms x %g)  %s %.3f (%.3f ms)" % (sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], r["kernel"], r["frac"], r["avg_launch_ms"], r["launches_per_step"],
      s.get("kernel", "-"), s.get("frac", 0), s.get("avg_launch_ms", 0)))
PY
  done
done
cat $OUT/summary.txt
