python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py tests/test_hip_edge.py -x -q 2>&1 | tail -4
for m in 0 1; do
RADMMM_DEBUG=1 RADMMM_MULTI_TRANSPOSE=$m python bench.py --step-only --steps 12 --warmup 4 2>&1 | tail -1
done
bash tools/prof_step.sh r04_m > /dev/null 2>&1; grep -a "weightnorm_fwd\|transpose_pair\|TOTAL" gpurun_out/r04_m_kernel_stats.txt | cut -c1-130
