set -e
mkdir -p gpurun_out/r2d
B="timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
FRAYHIP_PT_LANES=1 $B > gpurun_out/r2d/b1.json 2> gpurun_out/r2d/b1.err
FRAYHIP_LIB=build/ab/stamps/libfrayhip.so FRAYHIP_PT_LANES=1 timeout -k 10 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2d/stamps.json 2> gpurun_out/r2d/stamps.err
python - <<PY
import json
d=json.load(open("gpurun_out/r2d/b1.json"))
print(d["ms_per_step"], d["launch_ms_sums_per_step"])
PY
grep stamps gpurun_out/r2d/stamps.err
