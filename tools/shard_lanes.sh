#!/bin/bash
# Development aid (GPU box): one rank's share of an N-rank headline frame under different builds / numbers of batch lanes.   tools/shard_lanes.sh N "libs" lanes...
N=$1; LIBS=$2; shift 2
for l in $LIBS; do
  if [ "$l" = "default" ]; then unset FRAYHIP_LIB; else export FRAYHIP_LIB=$PWD/build/ab/$l/libfrayhip.so; fi
  for v in "$@"; do
    FRAYHIP_PT_LANES=$v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-serial-pass --shard-of $N --shard-rank 1 > gpurun_out/shard_${N}_$v.json 2> /dev/null
    python -c "
import json;d=json.load(open('gpurun_out/shard_${N}_$v.json'));print('$l: share of $N, lanes $v: %.3f ms' % d['ms_per_step'])"
  done
done
