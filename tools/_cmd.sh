python -m pytest tests/test_hip_round4.py -k "mas or ctc" -x -q 2>&1 | tail -5
