python -m pytest tests/test_conv_pk_gpu.py tests/test_conv_gpu.py tests/test_48k_gpu.py -m gpu -q 2>&1 | grep -E "^(FAILED)|passed|failed" | head -5
for e in "VCVITS_PK_NO_WS=1" "XX=1"; do echo "CFG $e"; env $e python tools/conv_layer_bench.py --reps 10 --only discP 2>/dev/null | cut -c1-100; done
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f32.json
