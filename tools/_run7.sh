set -e
mkdir -p gpurun_out/r2h
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_full_spp.py > gpurun_out/r2h/pytest.log 2>&1 || (tail -40 gpurun_out/r2h/pytest.log; exit 1)
tail -2 gpurun_out/r2h/pytest.log
for v in base sw5 bw4 bw2; do
  L=""; [ $v != base ] && L="build/ab/$v/libfrayhip.so"
  for lanes in 1 4; do
    FRAYHIP_LIB=$L FRAYHIP_PT_LANES=$lanes timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2h/${v}_$lanes.json 2> gpurun_out/r2h/${v}_$lanes.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2h/${v}_$lanes.json"))
print("$v lanes $lanes: %.2f ms" % d["ms_per_step"], {k:(round(x,1) if not isinstance(x,list) else x) for k,x in d["launch_ms_sums_per_step"].items()})
PY
  done
done
