#!/bin/bash
# What do the window kernel's epilogue and its stores cost?  Times the dominant launch (bench.py --dominant-only) on the
# product library and on two timing-only builds (results wrong):
#   RADMMM_KEEP_BUILDS=1 RADMMM_OUT=$PWD/rad_mmm_amd/libradmmm_hip_nostore.so bash rad_mmm_amd/csrc/build.sh -DRADMMM_EPI_NOSTORE
#   RADMMM_KEEP_BUILDS=1 RADMMM_OUT=$PWD/rad_mmm_amd/libradmmm_hip_noepi.so   bash rad_mmm_amd/csrc/build.sh -DRADMMM_EPI_NONE
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
for v in "" _nostore _noepi; do
  for rep in 1 2; do
    RADMMM_LIB_PATH="$ROOT/rad_mmm_amd/libradmmm_hip$v.so" python bench.py --dominant-only 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$v', round(d['avg_launch_ms']*1e3,1), 'us')"
  done
done
