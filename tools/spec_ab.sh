for v in 0 1; do
  FRAYHIP_SPECULATE_FANS=$v timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --workload dragon_whitted > gpurun_out/spec_$v.json 2> gpurun_out/spec_$v.err
  python -c "
import json;d=json.load(open('gpurun_out/spec_$v.json'));print('speculate=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['gpu_frame_vs_oracle_on_the_sample'] if 'gpu_frame_vs_oracle_on_the_sample' in d else '')"
done
