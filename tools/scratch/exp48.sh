python -m pytest tests/test_conv_gpu.py tests/test_modules_gpu.py -m gpu -q 2>&1 | grep -E "^(FAILED)|passed|failed" | head -4
python tools/conv_layer_bench.py --reps 20 --only "post" 2>/dev/null | cut -c1-100
python tools/conv_layer_bench.py --reps 20 --only "conv0" 2>/dev/null | cut -c1-100
