#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py tests/test_hip_round4.py tests/test_hip_round5.py tests/test_hip_lstm.py tests/test_hip_edge.py tests/test_attribute_predictors.py -m gpu -q -x 2>&1 | tail -3
(timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1 | cut -c1-150)
(timeout 600 python bench.py --config radmmm_splines --frames 2000 --steps 10 --warmup 3 --step-only 2>/dev/null | tail -1 | cut -c1-150)
