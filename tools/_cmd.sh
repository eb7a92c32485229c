#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_hip_round5.py -m gpu -q 2>&1 | tail -15)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 tests/_ddp_world2.py > gpurun_out/r05_w2.txt 2>&1
grep "DDP_WORLD2\|Error\|assert" gpurun_out/r05_w2.txt | head
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 tests/_ddp_world2.py --negative > gpurun_out/r05_w2n.txt 2>&1
grep "DDP_WORLD2\|Error\|assert" gpurun_out/r05_w2n.txt | head
for i in 1 2; do
(timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
(RADMMM_DEBUG=1 RADMMM_KEEP_FP32=1 timeout 600 python bench.py --steps 20 --warmup 5 --step-only 2>/dev/null | tail -1)
done
