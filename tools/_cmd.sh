#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export RADMMM_DEBUG=1
RADMMM_RES_SRC=0 timeout 600 bash tools/prof_step.sh res0 > /dev/null 2>&1
timeout 600 bash tools/prof_step.sh res1 > /dev/null 2>&1
for t in res0 res1; do echo "== $t"; grep -E "rowgemm_one_kernel|TOTAL" gpurun_out/${t}_kernel_stats.txt | cut -c1-130; done
