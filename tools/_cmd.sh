#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "cfg5" 2>&1 | grep -E "cfg5|passed|failed|Error" | tail -12 | cut -c1-400
