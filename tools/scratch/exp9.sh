python -m pytest tests/test_bf16_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error"
python tools/conv_layer_bench.py --dtype bf16 --reps 10 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12}'
python bench.py --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_bf16_voc.json
