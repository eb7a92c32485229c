#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export RADMMM_DEBUG=1
E=$PWD/rad_mmm_amd/libradmmm_hip_epi.so
(RADMMM_LIB_PATH=$E timeout 1200 python -m pytest tests/test_hip_round3.py tests/test_hip_round4.py tests/test_hip_round5.py -m gpu -q -x -k "not config5" 2>&1 | tail -3)
for i in 1 2; do
for v in "" _epi; do
echo -n "c5 lib$v: "; (RADMMM_LIB_PATH=$PWD/rad_mmm_amd/libradmmm_hip$v.so timeout 600 python bench.py --config radmmm_splines --frames 2000 --steps 10 --warmup 3 --step-only 2>/dev/null | tail -1 | cut -c1-120)
echo -n "c2 lib$v: "; (RADMMM_LIB_PATH=$PWD/rad_mmm_amd/libradmmm_hip$v.so timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1 | cut -c1-120)
done
done
