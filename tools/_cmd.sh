python -m pytest tests/test_ddp_nccl.py -x -q 2>&1 | tail -15
