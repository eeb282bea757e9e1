#!/bin/bash
# The measurement set kept under profiles/ for a round (run on the GPU box from the repo root).  A gpurun call is at most 20 minutes, so the set is taken in stages:
#   tools/final_runs.sh TAG profile "workload ..."     kernel trace + counters, serialised launches, for the named workloads -> gpurun_out/prof_TAG_<workload>/
#                                                      (tools/keep_profiles.sh TAG then puts them where bench.py looks: profiles/pmc_latest_<workload>.json)
#   tools/final_runs.sh TAG bench                      the bench lines (they carry `traffic` / `counters` when the profiles of the SAME source hash are in place)
#   tools/final_runs.sh TAG extra                      the 4096 x 4096 x 1024 spp frame, the headline under smaller queue budgets, every rank's share of an N-rank run
TAG=${1:-r04}; STAGE=${2:-bench}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
python tools/source_hash.py > $OUT/source_hash_$STAGE.txt
if [ "$STAGE" = "profile" ]; then
  for w in $3; do
    bash tools/profile_workload.sh ${TAG}_$w $w > $OUT/profile_$w.log 2>&1; echo "profile $w done" >> $OUT/progress.log
  done
elif [ "$STAGE" = "bench" ]; then
  # headline, default mode, with the CPU baseline leg
  timeout -k 10 600 python bench.py --steps 20 --warmup 2 > $OUT/bench_headline.json 2> $OUT/bench_headline.err; echo "headline done" >> $OUT/progress.log
  for w in ${BENCHED:-smallpt_pt64 boxed_whitted forest_dof16 forest_dof256 zaphod_whitted dragon_primary smallpt_whitted dragon_whitted bokeh_dof csg_nested_whitted csg_nested_pt16}; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $w > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "$w done" >> $OUT/progress.log
  done
elif [ "$STAGE" = "extra" ]; then
  timeout -k 10 400 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --workload smallpt_4k_pt1024 > $OUT/bench_smallpt_4k_pt1024.json 2> $OUT/bench_smallpt_4k_pt1024.err; echo "4k done" >> $OUT/progress.log
  for mib in 4096 8192; do
    FRAYHIP_PT_BUDGET_MIB=$mib timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_headline_budget_${mib}mib.json 2> /dev/null; echo "budget $mib done" >> $OUT/progress.log
  done
  timeout -k 10 300 python tools/shard_balance.py $OUT/shard_balance.json > $OUT/shard_balance.log 2>&1; echo "shard balance done" >> $OUT/progress.log
fi
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d["roofline"]
    print("%-44s %9.3f ms  %9.1f Mrays/s  roofline %s %.3f (%s %.3g %s)  hash %s" % (sys.argv[1].split("/")[-1], d["ms_per_step"], d["value"], r["bound"], r["frac"], r["kernel"], r["achieved"], r["unit"], d.get("source_hash")))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
