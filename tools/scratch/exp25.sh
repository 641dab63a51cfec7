python -m pytest tests/test_conv_gpu.py tests/test_round2_gpu.py tests/test_48k_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error" | head -20
for e in "WG_DBG=0" "WG_DBG=1"; do echo "CFG $e"; env $e python tools/conv_layer_bench.py --reps 10 2>/dev/null | cut -c1-100; done
