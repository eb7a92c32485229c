python bench.py --full-step 2> gpurun_out/r04_bench_e.err | tail -1 > gpurun_out/r04_bench_e.json
GRIDS="rowgemm_h3_kernel" bash tools/prof_full_step.sh r04_e > /dev/null 2>&1
python - <<PY
import json
d=json.loads(open('gpurun_out/r04_bench_e.json').read())
print(d['value'], d['ms_per_step_median']); print(d['parity_vs_cpu']); print(d['full_step']['ms_per_step'], d['full_step']['ms_outside_decoder_fwd_bwd'])
PY
