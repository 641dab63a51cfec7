python -m pytest tests/test_bf16_gpu.py tests/test_conv_pk_gpu.py tests/test_48k_gpu.py tests/test_conv_random_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error"
for dt in f32 bf16; do echo "== $dt"; python tools/conv_layer_bench.py --dtype $dt --reps 10 --only discP 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9}' | grep -E "conv[1-4]"; done
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_320.json
python bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_bf16_320.json
