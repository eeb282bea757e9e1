set -e
mkdir -p gpurun_out/r2e
for v in base nopool tw3 nopool_w5 sh4; do
  L=""; [ $v != base ] && L="build/ab/$v/libfrayhip.so"
  for lanes in 1 4; do
    FRAYHIP_LIB=$L FRAYHIP_PT_LANES=$lanes timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2e/${v}_$lanes.json 2> gpurun_out/r2e/${v}_$lanes.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2e/${v}_$lanes.json"))
print("$v lanes $lanes: %.2f ms" % d["ms_per_step"], {k:(round(x,1) if not isinstance(x,list) else x) for k,x in d["launch_ms_sums_per_step"].items()})
PY
  done
done
