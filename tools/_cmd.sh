#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_attribute_predictors.py tests/test_joint_step.py tests/test_hip_round6.py tests/test_tts_step.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -25
for i in 1 2; do
RADMMM_DEBUG=1 RADMMM_MERGED_LSTM_GEMMS=torch timeout 900 python bench.py --config joint --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('joint torch-gemms', d['full_step']['ms_per_step'])"
timeout 900 python bench.py --config joint --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('joint split-gemms', d['full_step']['ms_per_step'])"
done
