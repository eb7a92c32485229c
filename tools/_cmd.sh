#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export RADMMM_DEBUG=1
for i in 1 2 3; do
RADMMM_RES_STREAM=0 timeout 300 python bench.py --step-only --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-120 | sed 's/^/off  /' | tee -a gpurun_out/r06_b_ab.txt
RADMMM_RES_STREAM=1 timeout 300 python bench.py --step-only --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-120 | sed 's/^/on   /' | tee -a gpurun_out/r06_b_ab.txt
done
RADMMM_BENCH_SPAWN=1 timeout 300 python bench.py --step-only --steps 20 --warmup 5 2>&1 | grep step_only | cut -c1-120 | sed 's/^/pg on /' | tee -a gpurun_out/r06_b_ab.txt
RADMMM_BENCH_SPAWN=1 RADMMM_RES_STREAM=0 timeout 300 python bench.py --step-only --steps 20 --warmup 5 2>&1 | grep step_only | cut -c1-120 | sed 's/^/pg off/' | tee -a gpurun_out/r06_b_ab.txt
timeout 900 python -m pytest tests/test_hip_round6.py tests/test_hip_round5.py tests/test_tts_step.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r06_b_pytest.txt
