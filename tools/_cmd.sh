#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/pytest_gpu.txt 2>&1
tail -3 gpurun_out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --full-step > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['full_step']['ms_per_step'], d['parity_vs_cpu']['z_rel_err_vs_cpu'])
PY
timeout 900 python bench.py --config joint > gpurun_out/bench_joint.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_joint.json').read().strip().splitlines()[-1])
print('joint', d['value'], d['ms_per_step'])
PY
timeout 900 python bench.py --config radmmm_splines --frames 2000 > gpurun_out/bench_c5.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_c5.json').read().strip().splitlines()[-1])
print('c5', d['value'], d['ms_per_step'])
PY
