#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_hip_round3.py -m gpu -q -s -k "config5_defining" 2>&1 | grep -v "amdgpu.ids" | tail -30) > gpurun_out/r05_e_c5.txt
grep -n "configs\[4\] at\|knot accounting\|passed\|failed\|Error\|assert" gpurun_out/r05_e_c5.txt | head
(timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_edge.py tests/test_hip_round5.py tests/test_tts_step.py tests/test_attribute_predictors.py -m gpu -q -x 2>&1 | tail -8)
for i in 1 2; do
(timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
(RADMMM_DEBUG=1 RADMMM_COLSUM_BATCH=0 timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
done
