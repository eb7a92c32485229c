python bench.py --full-step --no-cpu-baseline --no-throughput-mode 2>&1 | tail -1 > gpurun_out/full_h.json
python - <<PY
import json
d=json.loads(open('gpurun_out/full_h.json').read())
print(d['ms_per_step_median']); print(json.dumps(d['full_step'],indent=1))
PY
