for m in 0 4; do for wl in zaphod_whitted forest_dof16 forest_dof256; do FRAYHIP_FUSED_WHITTED_MAX=$m timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/fz_${m}_$wl.json 2> gpurun_out/fz_${m}_$wl.err; python -c "
import json
r=json.loads([l for l in open('gpurun_out/fz_${m}_$wl.json') if l.startswith('{')][-1]); print('max $m $wl', r['ms_per_step'], r['roofline']['kernel'])"; done; done
FRAYHIP_FUSED_WHITTED_MAX=4 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_parity.py -x -q -k "whitted or fixture or fuzz or zaphod or full_size" > gpurun_out/fused_tests.log 2>&1; tail -n 2 gpurun_out/fused_tests.log
