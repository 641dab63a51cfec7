python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -E "^(FAILED|ERROR)|^tests.*(Error|FAIL)|rel err|^_____" | head -10
