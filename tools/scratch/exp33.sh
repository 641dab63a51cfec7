for w in 1 2 4 8; do echo "PK_WS=$w"; PK_WS=$w python -m pytest tests/test_conv_pk_gpu.py tests/test_conv_gpu.py -m gpu -q 2>&1 | grep -E "^(FAILED)|passed|failed" | head -8; done
