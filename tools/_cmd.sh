RADMMM_BENCH_SPAWN=1 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-throughput-mode 2>gpurun_out/spawn_err.txt | tail -1 > gpurun_out/spawn.json
tail -3 gpurun_out/spawn_err.txt
python - <<PY
import json
d=json.loads(open('gpurun_out/spawn.json').read())
print(d['ms_per_step_median']); print(json.dumps(d['distributed'])[:1800])
PY
python -m pytest tests/test_ddp_nccl.py -x -q 2>&1 | tail -2
