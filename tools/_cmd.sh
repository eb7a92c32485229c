#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/pytest_gpu.txt 2>&1
tail -2 gpurun_out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --full-step > gpurun_out/bench_e.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_e.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['full_step']['ms_per_step'])
PY
timeout 600 bash tools/prof_step.sh r05g > /dev/null 2>&1
head -6 gpurun_out/r05g_kernel_stats.txt | cut -c1-130
