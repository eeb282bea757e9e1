#!/bin/bash
# Copies the measurement set of tools/final_runs.sh TAG (merged back into gpurun_out/) to the tracked profiles/ directory -- and REFUSES a set whose files
# were not all measured on the device code of this tree (every pmc.json and every bench line carries the source hash, tools/source_hash.py).
#   tools/keep_profiles.sh TAG profiles     after the profile stages: pmc + kernel stats -> profiles/, pmc_latest_<workload>.json for the bench stage
#   tools/keep_profiles.sh TAG all          after the bench stages too
set -e
TAG=${1:-r04}; WHAT=${2:-all}
F=gpurun_out/final_$TAG
HASH=$(python3 tools/source_hash.py)
for P in gpurun_out/prof_${TAG}_*; do
  [ -f $P/pmc.json ] || continue
  w=${P#gpurun_out/prof_${TAG}_}
  h=$(python3 -c "import json; print(json.load(open('$P/pmc.json'))['source_hash'])")
  if [ "$h" != "$HASH" ]; then echo "REFUSED: $P/pmc.json was measured on source $h, the tree is $HASH"; exit 1; fi
  cp $P/pmc.json profiles/${TAG}_pmc_$w.json
  cp $P/pmc.json profiles/pmc_latest_$w.json
  cp $P/kernel_stats.csv profiles/${TAG}_kernel_stats_$w.csv
done
[ -f profiles/${TAG}_pmc_cornell_pt64.json ] && cp profiles/${TAG}_pmc_cornell_pt64.json profiles/pmc_latest.json
if [ "$WHAT" = "all" ]; then
  for f in $F/bench_*.json; do
    h=$(python3 -c "import json; print(json.load(open('$f')).get('source_hash'))")
    if [ "$h" != "$HASH" ]; then echo "REFUSED: $f was measured on source $h, the tree is $HASH"; exit 1; fi
    cp $f profiles/${TAG}_$(basename $f)
  done
  [ -f $F/shard_balance.json ] && cp $F/shard_balance.json profiles/${TAG}_shard_balance.json
fi
python3 tools/kernel_resources.py fray_amd/csrc/variant*.resources.txt > profiles/${TAG}_kernel_resources.txt
echo "kept: source hash $HASH"
