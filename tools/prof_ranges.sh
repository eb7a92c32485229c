#!/bin/bash
# Kernel time of the training step grouped by roctx range (flow<i>.fwd / flow<i>.bwd / context.fwd / lstm.bwd / loss):
# RADMMM_ROCTX=1 makes the host emit the ranges (rad_mmm_amd/_trace.py); rocprofv3 records them with the HIP launch calls and the
# dispatches; tools/kernel_stats.py --by-range joins the three.   usage (GPU box): bash tools/prof_ranges.sh <tag> [bench args...]
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="$1"; shift
OUT="$ROOT/gpurun_out/prof_ranges_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (rocprofv3 with --hip-trace has been seen to crash in its own teardown AFTER writing the database: the exit code is ignored)
RADMMM_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --hip-trace -d "$OUT" -- python "$ROOT/bench.py" --steps 3 --warmup 2 --step-only "$@" > "$OUT/bench.log" 2>&1 || true
DB=$(find "$OUT" -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2-)
cd "$ROOT"
python tools/kernel_stats.py "$DB" 5 --by-range > "gpurun_out/${TAG}_by_range.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -a step_only "$OUT/bench.log" | cut -c1-200
sed -n '/region categories/,$p' "gpurun_out/${TAG}_by_range.txt" | head -40
