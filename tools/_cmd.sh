#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python bench.py --step-only --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-110 | sed 's/^/default /' | tee -a gpurun_out/r06_pg_ab2.txt
RADMMM_BENCH_SPAWN=1 timeout 300 python bench.py --step-only --steps 20 --warmup 5 2>&1 | grep step_only | cut -c1-110 | sed 's/^/pg      /' | tee -a gpurun_out/r06_pg_ab2.txt
done
timeout 900 bash tools/pmc_dominant.sh gpurun_out/pmc_r06 r06 2>&1 | tail -2 | cut -c1-400
timeout 600 bash tools/prof_step.sh r06 > /dev/null 2>&1
head -8 gpurun_out/r06_kernel_stats.txt | cut -c1-130
