#!/bin/bash
# Copies the measurement set of tools/final_runs.sh TAG (merged back into gpurun_out/) to the tracked profiles/ directory.
set -e
TAG=${1:-r03}
F=gpurun_out/final_$TAG
for f in $F/bench_*.json; do cp $f profiles/${TAG}_$(basename $f); done
for w in cornell_pt64 forest_dof16 dragon_primary boxed_whitted dragon_whitted smallpt_pt64 zaphod_whitted smallpt_whitted bokeh_dof forest_dof256; do
  P=gpurun_out/prof_${TAG}_$w
  [ -f $P/pmc.json ] || continue
  cp $P/pmc.json profiles/${TAG}_pmc_$w.json
  cp $P/pmc.json profiles/pmc_latest_$w.json
  cp $P/kernel_stats.csv profiles/${TAG}_kernel_stats_$w.csv
done
cp profiles/${TAG}_pmc_cornell_pt64.json profiles/pmc_latest.json
cp $F/shard_balance.json profiles/${TAG}_shard_balance.json
make -s resources > /dev/null 2>&1 || true
python3 tools/kernel_resources.py fray_amd/csrc/variant*.resources.txt > profiles/${TAG}_kernel_resources.txt
echo "source hash of the tree: $(python3 tools/source_hash.py); of the profiles: $(python3 -c "import json; print(json.load(open('profiles/pmc_latest.json'))['source_hash'])")"
