set -e
mkdir -p gpurun_out/r2c
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "path_traced or full_size or batching or nested or fixture" > gpurun_out/r2c/pytest.log 2>&1 || (tail -40 gpurun_out/r2c/pytest.log; exit 1)
tail -2 gpurun_out/r2c/pytest.log
B="timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
$B > gpurun_out/r2c/b4.json 2> gpurun_out/r2c/b4.err
FRAYHIP_PT_LANES=1 $B > gpurun_out/r2c/b1.json 2> gpurun_out/r2c/b1.err
FRAYHIP_PT_BUDGET_MIB=8192 $B > gpurun_out/r2c/b4_8g.json 2> gpurun_out/r2c/b4_8g.err
$B --workload smallpt_pt64 > gpurun_out/r2c/smallpt.json 2> gpurun_out/r2c/smallpt.err
for f in b4 b1 b4_8g smallpt; do python - <<PY
import json
d=json.load(open("gpurun_out/r2c/$f.json"))
r=d["roofline"]; s=d.get("roofline_shadow_kernel",{})
print("$f", "ms/step %.2f" % d["ms_per_step"], "trace avg %.3f ms x %d" % (r["avg_launch_ms"], r["launches_per_step"]), "shadow avg %.3f" % s.get("avg_launch_ms",0), "kernels %.2f" % d["kernel_ms_per_step"])
PY
done
