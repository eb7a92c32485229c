#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_hip_round5.py -m gpu -q -x -k "independent" 2>&1 | tail -30
