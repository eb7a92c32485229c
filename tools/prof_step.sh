#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> per-kernel table with ms per step (tools/kernel_stats.py)
# usage (GPU box): bash tools/prof_step.sh <tag> [bench args...]   -> gpurun_out/<tag>_kernel_stats.txt
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="$1"; shift
OUT="$ROOT/gpurun_out/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -- python "$ROOT/bench.py" --steps 5 --warmup 2 --step-only "$@" > "$OUT/bench.log" 2>&1
# (a spawned run leaves one database per process: the rank's is the largest)
DB=$(find "$OUT" -name "*.db" -printf "%s %p\n" | sort -n | tail -1 | cut -d" " -f2-)
cd "$ROOT"
python tools/kernel_stats.py "$DB" 7 --gaps --hist rowgemm_win_kernelILi7ELi2 --json "gpurun_out/${TAG}_kernel_stats.json" > "gpurun_out/${TAG}_kernel_stats.txt" 2>&1
find "$OUT" -name "*.db" -delete
tail -1 "$OUT/bench.log" | cut -c1-300
head -45 "gpurun_out/${TAG}_kernel_stats.txt"
