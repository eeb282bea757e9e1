#!/bin/bash
for v in "$@"; do
  FRAYHIP_SEED_BLOCKS=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-serial-pass --workload forest_dof256 > gpurun_out/seed_$v.json 2> /dev/null
  python -c "
import json;d=json.load(open('gpurun_out/seed_$v.json'));print('seed blocks $v: %.3f ms' % d['ms_per_step'])"
done
