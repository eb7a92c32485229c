#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_joint_step.py -m gpu -q -s -k backward > gpurun_out/r06_g_joint.txt 2>&1; grep -n "kink\|compared\|^   [0-9]\|^{\|analytically\|passed\|failed" gpurun_out/r06_g_joint.txt | head -30
timeout 1500 python -m pytest tests/test_hip_edge.py tests/test_hip_aux.py tests/test_tts_step.py tests/test_hip_round6.py tests/test_ddp_nccl.py -m gpu -q 2>&1 | tail -5
