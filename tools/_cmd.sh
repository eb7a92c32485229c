#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/prof_step.sh r05 > gpurun_out/r05_prof.log 2>&1; tail -2 gpurun_out/r05_prof.log; head -12 gpurun_out/r05_kernel_stats.txt
bash tools/prof_step.sh r05_c5 --config radmmm_splines --frames 2000 > gpurun_out/r05_c5_prof.log 2>&1; tail -2 gpurun_out/r05_c5_prof.log; head -14 gpurun_out/r05_c5_kernel_stats.txt
PROBE_ARGS=--joint bash tools/prof_full_step.sh r05_joint > gpurun_out/r05_joint_prof.log 2>&1
head -16 gpurun_out/r05_joint_full_step_kernel_stats.txt
