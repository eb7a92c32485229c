#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export RADMMM_DEBUG=1
timeout 900 python -m pytest tests/test_hip_round5.py -m gpu -q -x -k "one_pass or repeatable or independent" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
(RADMMM_DACT_MULTI=0 timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
(timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
done
