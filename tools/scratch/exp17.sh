python -m pytest tests/test_conv_gpu.py tests/test_conv_random_gpu.py tests/test_48k_gpu.py tests/test_round2_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error"
python tools/conv_layer_bench.py --reps 10 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12}' | grep -E "res c|discP.*conv[1-4]|conv5"
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_w4.json
