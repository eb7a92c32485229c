python bench.py --no-cpu-baseline --no-throughput-mode 2>&1 | tail -1 > gpurun_out/bench_i.json
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_i.json').read())
print(d['ms_per_step_median']); r=d['roofline']; print({k:r[k] for k in ('achieved','frac','frac_executed','avg_launch_ms','avg_launch_measured','avg_launch_ms_back_to_back')})
PY
