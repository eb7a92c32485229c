#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python tools/lstm_volume_probe.py 2>&1 | tail -14 | tee gpurun_out/r06_lstm_volume.txt
timeout 600 python tools/lstm_volume_probe.py --frames 800 --hidden 128,384,524 2>&1 | tail -4 | tee -a gpurun_out/r06_lstm_volume.txt
