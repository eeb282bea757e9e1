#!/bin/bash
# Development aid (GPU box): a workload under different caps of k_seed's grid (FRAYHIP_SEED_BLOCKS), one line each.   tools/seed_sweep.sh WORKLOAD caps...
W=$1; shift
for v in "$@"; do
  FRAYHIP_SEED_BLOCKS=$v timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-serial-pass --workload $W > gpurun_out/seed_$v.json 2> /dev/null
  python -c "
import json;d=json.load(open('gpurun_out/seed_$v.json'));print('$W, seed blocks $v: %.3f ms' % d['ms_per_step'])"
done
