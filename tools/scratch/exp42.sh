python -m pytest tests/test_bf16_gpu.py tests/test_conv_pk_gpu.py -m gpu -q 2>&1 | grep -E "^(FAILED)|passed|failed" | head -4
python bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bf16.json
python bench.py --dtype bf16 --config 48k --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg5.json
