#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 bash tools/prof_step.sh r05f > /dev/null 2>&1
timeout 900 bash tools/prof_step.sh r05f_c5 --config radmmm_splines --frames 2000 > /dev/null 2>&1
PROBE_ARGS="--joint" timeout 900 bash tools/prof_full_step.sh r05f_joint > /dev/null 2>&1
head -8 gpurun_out/r05f_kernel_stats.txt | cut -c1-130
grep -a "step_only" gpurun_out/prof_r05f/bench.log gpurun_out/prof_r05f_c5/bench.log | cut -c1-200
