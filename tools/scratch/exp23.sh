python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error" | head -20
for d in 0 1; do echo "WG_DBG=$d"; WG_DBG=$d python tools/conv_layer_bench.py --reps 10 2>/dev/null | cut -c1-100; done
