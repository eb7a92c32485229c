#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ddp_nccl.py -m gpu -q -k "control_flow" 2>&1 | tail -3
for i in 1 2; do
timeout 900 python bench.py --config radmmm_splines --frames 2000 --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5', d['ms_per_step'], d['value'])"
done
timeout 300 python bench.py --step-only --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-120
