python -m pytest tests/test_hip_round4.py -k attention_core -x -q 2>&1 | tail -8
python -m pytest tests/test_hip_aux.py tests/test_tts_step.py -x -q -m gpu 2>&1 | tail -2
bash tools/prof_full_step.sh r04_h > /dev/null 2>&1; grep -a "attn_bwd" gpurun_out/r04_h_full_step_kernel_stats.txt
