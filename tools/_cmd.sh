#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
(timeout 2700 python -m pytest tests -m gpu -q --durations=15 2>&1 | grep -v "^\[Gloo\]\|amdgpu.ids" | tail -60) > gpurun_out/r05_pytest_gpu.txt
tail -30 gpurun_out/r05_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
