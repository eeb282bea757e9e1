#!/bin/bash
# Development aid (GPU box): the headline frame under different queue budgets (MiB) and builds, one line each.   tools/budget_sweep.sh "libs" budgets...
LIBS=$1; shift
for l in $LIBS; do
  if [ "$l" = "default" ]; then unset FRAYHIP_LIB; else export FRAYHIP_LIB=$PWD/build/ab/$l/libfrayhip.so; fi
  for mib in "$@"; do
    FRAYHIP_PT_BUDGET_MIB=$mib timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-serial-pass > gpurun_out/budget_$mib.json 2> /dev/null
    python -c "
import json;d=json.load(open('gpurun_out/budget_$mib.json'));print('$l, budget $mib MiB: %.3f ms' % d['ms_per_step'])"
  done
done
