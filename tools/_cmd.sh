python -m pytest tests/test_hip_aux.py tests/test_tts_step.py tests/test_hip_edge.py -x -q -m gpu 2>&1 | tail -2
GRIDS="rowgemm_h3d wgrad_h3 wgrad_rm_ rowgemm16 wgrad_f32 wgrad16 rowgemm_h3_kernel" bash tools/prof_full_step.sh r04_g 2>&1 | tail -1 | cut -c 200-330
