#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export RADMMM_DEBUG=1
for rep in 1 2; do
for v in "" _b128; do
for what in "1x1 res fwd" "1x1 plain"; do
RADMMM_LIB_PATH=$PWD/rad_mmm_amd/libradmmm_hip$v.so timeout 300 python tools/floor_probe.py --tag "lib${v:-_product}" --only "$what" 2>&1 | grep '^{' | cut -c1-150
done
done
done
