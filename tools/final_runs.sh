#!/bin/bash
# The measurement set kept under profiles/ for a round (run on the GPU box from the repo root): TAG = e.g. r03
TAG=${1:-r03}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
python tools/source_hash.py > $OUT/source_hash.txt
# 0. kernel trace + counters, serialised launches, for the workloads DESIGN.md quotes counters of -- FIRST, and put where bench.py looks
#    for them (profiles/pmc_latest_<workload>.json in this box's copy of the tree), so that the bench lines below carry `traffic` and `counters`
for w in ${PROFILED:-cornell_pt64 forest_dof16 dragon_primary boxed_whitted dragon_whitted smallpt_pt64 zaphod_whitted smallpt_whitted bokeh_dof forest_dof256}; do
  bash tools/profile_workload.sh ${TAG}_$w $w > $OUT/profile_$w.log 2>&1; cp gpurun_out/prof_${TAG}_$w/pmc.json profiles/pmc_latest_$w.json; echo "profile $w done" >> $OUT/progress.log
done
# 1. headline, default mode, with the CPU baseline leg
timeout -k 10 600 python bench.py --steps 20 --warmup 2 > $OUT/bench_headline.json 2> $OUT/bench_headline.err; echo "headline done" >> $OUT/progress.log
# 2. the other workloads (one GPU)
for w in ${BENCHED:-smallpt_pt64 boxed_whitted forest_dof16 forest_dof256 zaphod_whitted dragon_primary smallpt_whitted dragon_whitted bokeh_dof}; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w done" >> $OUT/progress.log
done
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-serial-pass --workload smallpt_4k_pt1024 > $OUT/bench_smallpt_4k_pt1024.json 2> $OUT/bench_smallpt_4k_pt1024.err; echo "4k done" >> $OUT/progress.log
# 3. headline under smaller queue budgets
for mib in 4096 8192; do
  FRAYHIP_PT_BUDGET_MIB=$mib timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_headline_budget_${mib}mib.json 2> /dev/null; echo "budget $mib done" >> $OUT/progress.log
done
# 5. every rank's share of an N-rank run, on this one GPU (a prediction of the compute side, not a scaling measurement)
timeout -k 10 300 python tools/shard_balance.py $OUT/shard_balance.json > $OUT/shard_balance.log 2>&1; echo "shard balance done" >> $OUT/progress.log
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%-44s %9.3f ms  %9.1f Mrays/s  roofline %s %.3f (%s %.3g %s)" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], r["bound"], r["frac"], r["kernel"], r["achieved"], r["unit"]))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
