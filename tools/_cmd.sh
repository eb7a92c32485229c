#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r06_pytest_gpu.txt 2>&1
tail -14 gpurun_out/r06_pytest_gpu.txt
timeout 900 python bench.py --config joint --no-throughput-mode > gpurun_out/r06_bench_joint.json 2> gpurun_out/r06_bench_joint.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_joint.json').read().strip().splitlines()[-1])
f=d['full_step']
print('joint', f['ms_per_step'], f.get('host_syncs_per_step'), json.dumps(f.get('parity_vs_cpu'))[:400])
PY
PROBE_ARGS=--joint timeout 900 bash tools/prof_full_step.sh r06_joint > /dev/null 2>&1
head -30 gpurun_out/r06_joint_full_step_kernel_stats.txt | cut -c1-140
