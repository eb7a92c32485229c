#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 bash tools/prof_ranges.sh r06 2>&1 | tail -40
