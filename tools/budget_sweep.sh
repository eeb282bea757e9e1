#!/bin/bash
# Development aid (GPU box): the headline frame under different queue budgets (MiB), one line each.
for mib in "$@"; do
  FRAYHIP_PT_BUDGET_MIB=$mib timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-serial-pass > gpurun_out/budget_$mib.json 2> /dev/null
  python -c "
import json;d=json.load(open('gpurun_out/budget_$mib.json'));print('budget $mib MiB: %.3f ms' % d['ms_per_step'], d['config'].get('batches'), d['config'].get('spp_per_batch'))"
done
