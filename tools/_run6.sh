set -e
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_full_spp.py > gpurun_out/r2g/pytest.log 2>&1 || (tail -40 gpurun_out/r2g/pytest.log; exit 1)
tail -2 gpurun_out/r2g/pytest.log
for v in base; do
  for lanes in 1 4; do
    FRAYHIP_PT_LANES=$lanes timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2g/${v}_$lanes.json 2> gpurun_out/r2g/${v}_$lanes.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r2g/${v}_$lanes.json"))
print("$v lanes $lanes: %.2f ms" % d["ms_per_step"], {k:(round(x,1) if not isinstance(x,list) else x) for k,x in d["launch_ms_sums_per_step"].items()})
PY
  done
done
for w in smallpt_pt64 boxed_whitted forest_dof16 dragon_primary zaphod_whitted; do
  timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $w > gpurun_out/r2g/$w.json 2> gpurun_out/r2g/$w.err
  python -c "
import json; d=json.load(open('gpurun_out/r2g/$w.json')); print('$w %.3f ms' % d['ms_per_step'])"
done
FRAYHIP_LIB=build/ab/stamps/libfrayhip.so FRAYHIP_PT_LANES=1 timeout -k 10 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r2g/stamps.json 2> gpurun_out/r2g/stamps.err
grep stamps gpurun_out/r2g/stamps.err
