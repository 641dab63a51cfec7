PK_WS8=1 python -m pytest tests/test_conv_pk_gpu.py tests/test_conv_gpu.py -m gpu -q 2>&1 | grep -E "^(FAILED)|passed|failed" | head -5
for e in "XX=1" "PK_WS8=1"; do echo "CFG $e"; env $e python tools/conv_layer_bench.py --reps 10 --only discP 2>/dev/null | cut -c1-100; done
