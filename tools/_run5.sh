set -e
mkdir -p gpurun_out/r2f
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "path_traced or full_size_prop or batching" > gpurun_out/r2f/pytest.log 2>&1 || (tail -40 gpurun_out/r2f/pytest.log; exit 1)
tail -2 gpurun_out/r2f/pytest.log
for v in base nopool pb2 pb4; do
  L=""; [ $v != base ] && L="build/ab/$v/libfrayhip.so"
  for lanes in 1 4; do
    FRAYHIP_LIB=$L FRAYHIP_PT_LANES=$lanes timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2f/${v}_$lanes.json 2> gpurun_out/r2f/${v}_$lanes.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2f/${v}_$lanes.json"))
print("$v lanes $lanes: %.2f ms" % d["ms_per_step"], {k:(round(x,1) if not isinstance(x,list) else x) for k,x in d["launch_ms_sums_per_step"].items()})
PY
  done
done
FRAYHIP_LIB=build/ab/stamps/libfrayhip.so FRAYHIP_PT_LANES=1 timeout -k 10 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2f/stamps.json 2> gpurun_out/r2f/stamps.err
grep stamps gpurun_out/r2f/stamps.err
