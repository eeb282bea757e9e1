mkdir -p gpurun_out/r2l
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "primary or whitted or path_traced" > gpurun_out/r2l/pytest.log 2>&1; tail -3 gpurun_out/r2l/pytest.log
for v in base kdtop0 kdtop1020; do
L=""; [ $v != base ] && L="build/ab/$v/libfrayhip.so"
for w in dragon_primary boxed_whitted forest_dof16 cornell_pt64; do
  FRAYHIP_LIB=$L timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $w > gpurun_out/r2l/${v}_$w.json 2> gpurun_out/r2l/${v}_$w.err
  python -c "
import json; d=json.load(open('gpurun_out/r2l/${v}_$w.json')); print('$v $w %.3f ms' % d['ms_per_step'], {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['launch_ms_sums_per_step'].items() if k!='note'})" || tail -5 gpurun_out/r2l/${v}_$w.err
done
done
