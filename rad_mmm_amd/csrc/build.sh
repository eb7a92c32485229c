#!/bin/bash
# Build libradmmm_hip.so in-tree for gfx950 (cross-compiles without a GPU).
# Every source is compiled to its own object (in parallel, rebuilt only when it or a header is
# newer) under build/ and the objects are linked into one shared library.  Extra arguments are
# passed to every compile (e.g. -DRADMMM_ABLATION); they are part of the cache key.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${RADMMM_OUT:-$HERE/../libradmmm_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@")
KEY="$(printf '%s ' "${FLAGS[@]}" | md5sum | cut -c1-8)"
OBJ="$HERE/build/$KEY"
mkdir -p "$OBJ"
NEWEST_HDR="$(ls -t "$HERE"/*.h "$HERE/../../include"/*.h | head -1)"
pids=()
objs=()
for src in "$HERE"/*.hip "$HERE"/error.cpp; do
  o="$OBJ/$(basename "$src").o"
  objs+=("$o")
  if [[ ! -f "$o" || "$src" -nt "$o" || "$NEWEST_HDR" -nt "$o" ]]; then
    ( "$HIPCC" "${FLAGS[@]}" -c "$src" -o "$o.tmp" && mv "$o.tmp" "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do
  [[ -n "$p" ]] && wait "$p"
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT.tmp"      # only objects of sources that still exist
mv "$OUT.tmp" "$OUT"
# housekeeping: objects of sources that no longer exist, and the object caches of other flag sets (each ~3 MB that would
# travel to the GPU box with every run); RADMMM_KEEP_BUILDS=1 keeps the latter (A/B builds with extra -D flags)
for o in "$OBJ"/*.o; do
  [[ -f "$HERE/$(basename "${o%.o}")" ]] || rm -f "$o"
done
# Only caches nothing has touched for two hours are removed: a concurrent build with other flags (an A/B build with -D
# options next to pytest's default build) keeps its directory while it compiles and links.
if [[ "${RADMMM_KEEP_BUILDS:-0}" != "1" ]]; then
  for d in "$HERE"/build/*/; do
    [[ -d "$d" && "$(basename "$d")" != "$KEY" ]] || continue
    [[ -n "$(find "$d" -maxdepth 1 -mmin -120 -print -quit)" ]] || rm -rf "$d"
  done
fi
echo "built $OUT"
