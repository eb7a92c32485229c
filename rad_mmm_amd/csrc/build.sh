#!/bin/bash
# Build libradmmm_hip.so in-tree for gfx950 (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libradmmm_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS=("$HERE"/*.hip "$HERE"/error.cpp)
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function)
"$HIPCC" "${FLAGS[@]}" "${SRCS[@]}" -o "$OUT" "$@"
echo "built $OUT"
