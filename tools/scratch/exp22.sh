python -m pytest tests/test_conv_gpu.py tests/test_conv_pk_gpu.py tests/test_round2_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error" | head -20
python tools/conv_layer_bench.py --reps 10 2>/dev/null > gpurun_out/layers_wg.txt
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_wg.json
