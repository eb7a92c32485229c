#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r06_pytest_gpu_final.txt 2>&1
tail -12 gpurun_out/r06_pytest_gpu_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['parity_vs_cpu']['z_rel_err_vs_cpu'], d['parity_vs_cpu']['nll_rel_diff_vs_cpu'], d['cpu_baseline']['value'])
print([ (k['kernel'][:28], round(k['frac'],3)) for k in d['roofline_mfma']['kernels']], d['process_group_overhead_ms']['value'])
PY
timeout 900 python bench.py --config joint --no-throughput-mode > gpurun_out/r06_bench_joint_final.json 2>/dev/null
timeout 900 python bench.py --config radmmm_splines --frames 2000 --no-throughput-mode --no-cpu-baseline > gpurun_out/r06_bench_splines_final.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_bench_joint_final.json').read().strip().splitlines()[-1]); print('joint', d['full_step']['ms_per_step'], d['ms_per_step'])
d=json.loads(open('gpurun_out/r06_bench_splines_final.json').read().strip().splitlines()[-1]); print('c5', d['ms_per_step'], d['value'])
PY
timeout 600 bash tools/prof_step.sh r06_final > /dev/null 2>&1
head -5 gpurun_out/r06_final_kernel_stats.txt | cut -c1-130
