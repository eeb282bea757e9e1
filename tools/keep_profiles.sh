#!/bin/bash
# Copies the measurement set of tools/final_runs.sh TAG (merged back into gpurun_out/) to the tracked profiles/ directory.
set -e
TAG=${1:-r02}
F=gpurun_out/final_$TAG
for f in $F/bench_*.json; do cp $f profiles/${TAG}_$(basename $f); done
cp gpurun_out/prof_$TAG/pmc.json profiles/${TAG}_pmc.json
cp gpurun_out/prof_$TAG/pmc.json profiles/pmc_latest.json
cp gpurun_out/prof_${TAG}_forest/pmc.json profiles/${TAG}_pmc_forest_dof16.json
cp "$(ls -t gpurun_out/prof_$TAG/trace/*/*_kernel_stats.csv | head -1)" profiles/${TAG}_kernel_stats.csv
cp "$(ls -t gpurun_out/prof_${TAG}_forest/trace/*/*_kernel_stats.csv | head -1)" profiles/${TAG}_kernel_stats_forest_dof16.csv
make -s resources > /dev/null 2>&1 || true
python3 tools/kernel_resources.py fray_amd/csrc/variant*.resources.txt > profiles/${TAG}_kernel_resources.txt
echo "source hash of the tree: $(python3 tools/source_hash.py); of the profile: $(python3 -c "import json; print(json.load(open('profiles/pmc_latest.json'))['source_hash'])")"
