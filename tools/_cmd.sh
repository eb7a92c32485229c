( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r04_pytest_gpu.txt 2>&1
tail -4 gpurun_out/r04_pytest_gpu.txt
python bench.py --full-step 2> gpurun_out/r04_bench_f.err | tail -1 > gpurun_out/r04_bench_f.json
bash tools/prof_step.sh r04_f > /dev/null 2>&1
GRIDS="rowgemm_h3_kernel" bash tools/prof_full_step.sh r04_f > /dev/null 2>&1
python - <<PY
import json
d=json.loads(open('gpurun_out/r04_bench_f.json').read())
print(d['value'], d['ms_per_step_median'], d['roofline']['frac'], d['parity_vs_cpu']['z_rel_err_vs_cpu'], d['full_step']['ms_per_step'], d['full_step']['ms_outside_decoder_fwd_bwd'])
PY
