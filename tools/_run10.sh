mkdir -p gpurun_out/r2k
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "primary or whitted" > gpurun_out/r2k/pytest.log 2>&1; tail -3 gpurun_out/r2k/pytest.log
for w in dragon_primary boxed_whitted forest_dof16; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $w > gpurun_out/r2k/$w.json 2> gpurun_out/r2k/$w.err
  python -c "
import json; d=json.load(open('gpurun_out/r2k/$w.json')); print('$w %.3f ms' % d['ms_per_step'], {k:v for k,v in d['launch_ms_sums_per_step'].items() if k!='note'})" || tail -5 gpurun_out/r2k/$w.err
done
