#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_attribute_predictors.py tests/test_joint_step.py tests/test_tts_step.py -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" | grep "merged vs\|DAP B\|passed\|failed\|Error\|summed_loss\|assert" | head -20)
(timeout 900 python bench.py --config joint --steps 10 --warmup 3 --no-throughput-mode 2>gpurun_out/r05_f_joint.err | tail -1) > gpurun_out/r05_bench_joint.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_joint.json'))
f=d['full_step']
print('decoder-only ms', d['ms_per_step'], 'joint ms', f['ms_per_step'], 'value', f['value'], 'syncs', f['host_syncs_per_step'])
print(f['split'])
PY
(timeout 900 python bench.py --full-step --steps 10 --warmup 3 --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r05_bench_full_step.json
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_full_step.json')); f=d['full_step']; print('full step ms', f['ms_per_step'], 'outside decoder', f['ms_outside_decoder_fwd_bwd'], 'syncs', f['host_syncs_per_step'], f['mas_alignments_changed_by_log_choice']['differing'])"
PROBE_ARGS=--joint bash tools/prof_full_step.sh r05_joint > gpurun_out/r05_joint_prof.log 2>&1
head -24 gpurun_out/r05_joint_full_step_kernel_stats.txt; grep -n "idle" gpurun_out/r05_joint_full_step_kernel_stats.txt
