set -e
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1 || (tail -30 gpurun_out/r2a/pytest.log; exit 1)
tail -2 gpurun_out/r2a/pytest.log
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
$B > gpurun_out/r2a/base4.json 2> gpurun_out/r2a/base4.err
FRAYHIP_PT_LANES=1 $B > gpurun_out/r2a/base1.json 2> gpurun_out/r2a/base1.err
FRAYHIP_PT_BUDGET_MIB=8192 $B > gpurun_out/r2a/base4_8g.json 2> gpurun_out/r2a/base4_8g.err
for v in notri alltri; do
  FRAYHIP_LIB=build/ab/$v/libfrayhip.so FRAYHIP_PT_LANES=1 $B > gpurun_out/r2a/${v}1.json 2> gpurun_out/r2a/${v}1.err
  FRAYHIP_LIB=build/ab/$v/libfrayhip.so $B > gpurun_out/r2a/${v}4.json 2> gpurun_out/r2a/${v}4.err
done
FRAYHIP_LIB=build/ab/stamps/libfrayhip.so FRAYHIP_PT_LANES=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2a/stamps.json 2> gpurun_out/r2a/stamps.err
for f in base4 base1 base4_8g notri1 notri4 alltri1 alltri4 stamps; do python - <<PY
import json
d=json.load(open("gpurun_out/r2a/$f.json"))
r=d["roofline"]; s=d.get("roofline_shadow_kernel",{})
print("$f", "ms/step %.2f" % d["ms_per_step"], "bounce avg %.3f ms x %d" % (r["avg_launch_ms"], r["launches_per_step"]), "shadow avg %.3f" % s.get("avg_launch_ms",0), "kernels %.2f" % d["kernel_ms_per_step"])
PY
done
tail -40 gpurun_out/r2a/stamps.err
