#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in "" _skiprd; do
RADMMM_LIB_PATH=$PWD/rad_mmm_amd/libradmmm_hip$v.so python tools/floor_probe.py --tag "lib${v:-_product}" --only wgrad 2>&1 | grep '^{'
RADMMM_LIB_PATH=$PWD/rad_mmm_amd/libradmmm_hip$v.so python tools/floor_probe.py --tag "lib${v:-_product}" --only wgrad 2>&1 | grep '^{'
done
(timeout 900 python -m pytest tests/test_attribute_predictors.py tests/test_encoder.py tests/test_tts_step.py tests/test_hip_aux.py -m gpu -q 2>&1 | tail -3)
(timeout 900 python bench.py --config joint --steps 10 --warmup 3 --no-throughput-mode --no-cpu-baseline 2>/dev/null | tail -1) | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['full_step']; print('joint ms', f['ms_per_step'], 'decoder', d['ms_per_step'], 'syncs', f['host_syncs_per_step'])"
