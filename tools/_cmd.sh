#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ddp_nccl.py tests/test_tts_step.py tests/test_hip_edge.py tests/test_radam.py tests/test_hip_round5.py tests/test_joint_step.py tests/test_hip_round4.py -m gpu -q -x 2>&1 | tail -8
for i in 1 2; do
timeout 900 python bench.py --config joint --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('joint', d['full_step']['ms_per_step'], 'decoder', d['ms_per_step'])"
done
timeout 300 python bench.py --step-only --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-120
RADMMM_BENCH_SPAWN=1 timeout 300 python bench.py --step-only --steps 20 --warmup 5 2>&1 | grep step_only | cut -c1-120
timeout 600 python bench.py --full-step --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('full_step', d['full_step']['ms_per_step'], d['full_step'].get('host_syncs_per_step'))"
