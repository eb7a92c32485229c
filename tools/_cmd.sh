#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
RADMMM_TEXT_WGRAD8=1 timeout 1500 python -m pytest tests/test_joint_step.py tests/test_tts_step.py tests/test_encoder.py tests/test_hip_aux.py -m gpu -q -s 2>&1 | grep -E "passed|failed|Error|^\{|compared|^   [0-9]" | tail -16 | cut -c1-300
for i in 1 2; do
RADMMM_TEXT_WGRAD8=0 timeout 900 python bench.py --config joint --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('joint text_wgrad8=0', d['full_step']['ms_per_step'])"
RADMMM_TEXT_WGRAD8=1 timeout 900 python bench.py --config joint --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('joint text_wgrad8=1', d['full_step']['ms_per_step'])"
done
