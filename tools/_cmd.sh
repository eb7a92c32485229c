python -m pytest tests/test_hip_round4.py -k ctc -x -q 2>&1 | tail -1
python bench.py --full-step --no-cpu-baseline --no-throughput-mode 2>&1 | tail -1 > gpurun_out/full_e.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/full_e.json').read())
print(d['ms_per_step'], d.get('ms_per_step_median'))
print(json.dumps(d['full_step'], indent=1))
PY
