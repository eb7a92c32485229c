#!/bin/bash
# rocprofv3 kernel trace of the full training step (tools/full_step_probe.py) -> gpurun_out/<tag>_full_step_kernel_stats.txt
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="$1"; shift
OUT="$ROOT/gpurun_out/prof_full_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -- python "$ROOT/tools/full_step_probe.py" --steps 7 ${PROBE_ARGS:-} > "$OUT/bench.log" 2>&1
DB=$(find "$OUT" -name "*.db" | head -1)
cd "$ROOT"
python tools/kernel_stats.py "$DB" 12 --gaps --json "gpurun_out/${TAG}_full_step_kernel_stats.json" > "gpurun_out/${TAG}_full_step_kernel_stats.txt" 2>&1
python tools/kernel_grids.py "$DB" ${GRIDS:-rowgemm_h3d_kernelILi8ELi3ELi1 rowgemm16 wgrad_f32 wgrad16} > "gpurun_out/${TAG}_full_step_grids.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -a what "$OUT/bench.log" | cut -c1-600
