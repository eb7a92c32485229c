#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export RADMMM_DEBUG=1
L=$PWD/rad_mmm_amd/libradmmm_hip_bglobal.so
(RADMMM_LIB_PATH=$L timeout 900 python -m pytest tests/test_hip_round3.py -m gpu -q -k "shared_window" 2>&1 | tail -4)
for v in "" _bglobal; do
RADMMM_LIB_PATH=$PWD/rad_mmm_amd/libradmmm_hip$v.so python tools/floor_probe.py --tag "lib${v:-_product}" --only "5-tap fwd, SPLIT epilogue, pair only" 2>&1 | grep '^{'
RADMMM_LIB_PATH=$PWD/rad_mmm_amd/libradmmm_hip$v.so python tools/floor_probe.py --tag "lib${v:-_product}" --only "fused dgrad, dact from the split pair, pair only" 2>&1 | grep '^{'
done
for i in 1 2; do
(timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
(RADMMM_LIB_PATH=$L timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
done
