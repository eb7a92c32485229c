#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export RADMMM_DEBUG=1
RADMMM_CONVNORM_H3_MIN_ROWS=100000000 timeout 1500 python -m pytest tests/test_joint_step.py -m gpu -q -s -k backward > gpurun_out/r06_f_joint_fp32conv.txt 2>&1; echo "--- conv_norm on fp32 kernels"; grep -n "kink\|compared\|^   [0-9]\|^{" gpurun_out/r06_f_joint_fp32conv.txt | head -16
RADMMM_CONV_OWN_SCALE=1 timeout 1500 python -m pytest tests/test_joint_step.py -m gpu -q -s -k backward > gpurun_out/r06_f_joint_ownscale.txt 2>&1; echo "--- every conv its own gradient scale"; grep -n "kink\|compared\|^   [0-9]\|^{" gpurun_out/r06_f_joint_ownscale.txt | head -16
