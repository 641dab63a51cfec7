PK_WS_BF16=1 python -m pytest tests/test_bf16_gpu.py -m gpu -q 2>&1 | grep -E "^(FAILED)|passed|failed" | head -5
for e in "XX=1" "PK_WS_BF16=1"; do echo "CFG $e"; env $e python tools/conv_layer_bench.py --reps 10 --dtype bf16 2>/dev/null | cut -c1-100; env $e python bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bf16_$e.json; done
