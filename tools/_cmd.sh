python -m pytest tests/test_hip_round4.py -k ctc -x -q -s 2>&1 | grep -v Warn | tail -14
python -m pytest tests/test_hip_aux.py tests/test_tts_step.py -x -q -m gpu 2>&1 | tail -2
bash tools/prof_full_step.sh r04_h > /dev/null 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/r04_h_full_step_kernel_stats.json'))
for n,k in d['kernels'].items():
    if 'ctc' in n or 'mas' in n: print(n[:70], k['calls'], round(k['avg_us'],1))
PY
