for t in 0 1; do echo "== tall $t"; VCVITS_PK_TALL=$t python tools/conv_layer_bench.py --reps 10 --only discP 2>&1 | grep -E "conv[3-4]" | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9}'; done
VCVITS_PK_TALL=1 python -m pytest tests/test_48k_gpu.py tests/test_conv_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error"
VCVITS_PK_TALL=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_tall.json
