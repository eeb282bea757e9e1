#!/bin/bash
# Development aid (GPU box): one rank's share of an N-rank headline frame under different queue budgets (MiB).   tools/shard_budget.sh N budgets...
N=$1; shift
for mib in "$@"; do
  FRAYHIP_PT_BUDGET_MIB=$mib timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-serial-pass --shard-of $N --shard-rank 1 > gpurun_out/shardb_$mib.json 2> /dev/null
  python -c "
import json;d=json.load(open('gpurun_out/shardb_$mib.json'));print('share of $N, budget $mib MiB: %.3f ms' % d['ms_per_step'])"
done
