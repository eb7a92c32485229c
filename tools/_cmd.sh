python -m pytest tests/test_radam.py tests/test_hip_edge.py tests/test_tts_step.py tests/test_ddp_nccl.py -x -q -m gpu 2>&1 | tail -3
python bench.py --step-only --steps 12 --warmup 4 2>&1 | tail -1
bash tools/prof_step.sh r04_i > /dev/null 2>&1; grep -a "weightnorm_bwd" gpurun_out/r04_i_kernel_stats.txt | cut -c1-130
