#!/bin/bash
# Development aid (GPU box): ms per frame of workloads on several builds (tools/build_ab.sh names, or "default"), alternating, two rounds.
#   tools/ab_quick.sh "lib1 lib2" workload ...
LIBS=$1; shift
for rep in 1 2; do for w in "$@"; do for l in $LIBS; do
  if [ "$l" = "default" ]; then unset FRAYHIP_LIB; else export FRAYHIP_LIB=$PWD/build/ab/$l/libfrayhip.so; fi
  timeout -k 10 200 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-contracted 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s %-18s %9.3f ms' % ('$l', '$w', r['ms_per_step']))"
done; done; done
