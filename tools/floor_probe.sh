#!/bin/bash
# Non-MFMA floor of the GEMM launches and the weight gradient's conversion cost (profiles/r05_nprod1_floor.txt,
# profiles/r05_wgrad_hi8.txt).  Build the timing-only libraries HERE first (hipcc cross-compiles; ~45 s each):
#   for v in 1 2 3; do RADMMM_KEEP_BUILDS=1 RADMMM_OUT=$PWD/rad_mmm_amd/libradmmm_hip_t$v.so bash rad_mmm_amd/csrc/build.sh -DRADMMM_TIMING=$v; done
#   RADMMM_KEEP_BUILDS=1 RADMMM_OUT=$PWD/rad_mmm_amd/libradmmm_hip_nocvt.so bash rad_mmm_amd/csrc/build.sh -DRADMMM_TIMING_NOCVT
# then on the GPU box:  bash tools/floor_probe.sh > gpurun_out/floor_probe.txt
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
for v in "" _t1 _t2 _t3 _nocvt; do
  lib="$ROOT/rad_mmm_amd/libradmmm_hip$v.so"
  [[ -f "$lib" ]] || continue
  only=""
  [[ "$v" == "_nocvt" ]] && only="--only wgrad"
  RADMMM_LIB_PATH="$lib" python tools/floor_probe.py --tag "lib${v:-_product}" $only 2>&1 | grep '^{'
done
