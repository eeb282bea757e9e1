#!/bin/bash
# Development aid (GPU box): is a round-to-round difference the library's or the harness's?  Round 4's tree with its own library, round 4's tree
# (its bench.py, its Python package) with THIS tree's library, and this tree, alternating on one box.
W=${1:-smallpt_pt64}; ROOT=$(pwd)
for rep in 1 2 3; do
  (cd build/r4tree && timeout -k 10 200 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r4 harness, r4 library  %-16s %9.3f ms' % ('$W', r['ms_per_step']))")
  (cd build/r4tree && FRAYHIP_LIB=$ROOT/fray_amd/libfrayhip.so timeout -k 10 200 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r4 harness, r5 library  %-16s %9.3f ms' % ('$W', r['ms_per_step']))")
  timeout -k 10 200 python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-serial-pass --no-contracted 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r5 harness, r5 library  %-16s %9.3f ms' % ('$W', r['ms_per_step']))"
done
