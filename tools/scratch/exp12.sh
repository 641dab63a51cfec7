VCVITS_CONV_PK=1 python -m pytest tests/test_conv_gpu.py tests/test_conv_random_gpu.py tests/test_48k_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error"
for pk in 0 1; do echo "== pk $pk"; VCVITS_CONV_PK=$pk python tools/conv_layer_bench.py --reps 10 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12}'; done
