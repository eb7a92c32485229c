#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export RADMMM_DEBUG=1
E=$PWD/rad_mmm_amd/libradmmm_hip_epi.so
(RADMMM_LIB_PATH=$E timeout 1200 python -m pytest tests/test_hip_round3.py tests/test_hip_round4.py tests/test_hip_round5.py -m gpu -q -x -k "not config5" 2>&1 | tail -3)
for i in 1 2 3; do
(timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
(RADMMM_LIB_PATH=$E timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
done
