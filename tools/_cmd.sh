#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 bash tools/prof_step.sh mix > /dev/null 2>&1
grep -E "rowgemm_mix|rowgemm16|wgrad16|wn_input" gpurun_out/mix_kernel_stats.txt | cut -c1-130
