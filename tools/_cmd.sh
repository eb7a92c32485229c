#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 400 python tools/aten_ops_probe.py radmmm_splines 2000 2>&1 | tail -60 > gpurun_out/aten_c5.txt
timeout 300 python tools/aten_ops_probe.py radtts 800 2>&1 | tail -50 > gpurun_out/aten_c2.txt
head -50 gpurun_out/aten_c5.txt
