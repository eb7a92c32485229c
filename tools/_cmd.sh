#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
(timeout 900 python bench.py --full-step 2>gpurun_out/r05_bench_a.err | tail -1) > gpurun_out/r05_bench_a.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_a.json'))
print('default ms', d['ms_per_step'], d['value'], 'roofline', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], 'parity', d['parity_vs_cpu']['z_rel_err_vs_cpu'], d['parity_vs_cpu']['nll_rel_diff_vs_cpu'], 'cpu', d['cpu_baseline']['value'], 'h3', d['exact_split_mode']['ms_per_step'], 'f16', d['throughput_mode']['ms_per_step'])
f=d['full_step']; print('full step', f['ms_per_step'], f['host_syncs_per_step'], f['mas_alignments_changed_by_log_choice'])
print(d['saturation'])
print([ (r['kernel'], round(r['frac'],2)) for r in d['roofline_hbm']['kernels']] if 'kernels' in d['roofline_hbm'] else d['roofline_hbm'].keys())
PY
(timeout 900 python bench.py --config joint --no-throughput-mode 2>/dev/null | tail -1) > gpurun_out/r05_bench_joint.json
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_joint.json')); f=d['full_step']; print('joint', f['ms_per_step'], f['value'], f['host_syncs_per_step'], f['parity_vs_cpu']['summed_loss_rel_diff'], f['parity_vs_cpu']['predictor_output_rel_err_vs_cpu'])"
(timeout 900 python bench.py --config radmmm_splines --frames 2000 --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r05_bench_splines_T2000.json
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_splines_T2000.json')); print('splines', d['ms_per_step'], d['value'])"
(timeout 900 python bench.py --config radmmm --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r05_bench_radmmm.json
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_radmmm.json')); print('radmmm', d['ms_per_step'], d['value'])"
(RADMMM_BENCH_SPAWN=1 timeout 900 python bench.py --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r05_bench_rccl_world1.json
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_rccl_world1.json')); print('rccl world1', d['ms_per_step'], d['distributed']['exposed_comm_ms_median_rank0'], d['distributed']['gradient_buckets'])"
