#!/bin/bash
# Development aid (GPU box): the same workloads on round 4's tree (build/r4tree, built from `git archive d00ef42`) and on this tree, alternating, same box.
for rep in 1 2; do
for w in "$@"; do
  (cd build/r4tree && timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r4  %-18s %9.3f ms' % ('$w', r['ms_per_step']))")
  timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-contracted 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r5  %-18s %9.3f ms' % ('$w', r['ms_per_step']))"
done
done
