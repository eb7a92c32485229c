#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python bench.py --full-step > gpurun_out/bench_d.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_d.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['full_step']['ms_per_step'])
PY
