for hm in 0 1; do echo "== halfm $hm"; VCVITS_PK_HALFM=$hm python tools/conv_layer_bench.py --reps 10 --only discP 2>&1 | grep -v amdgpu.ids | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9}' | grep -E "conv[1-4]"; done
VCVITS_PK_HALFM=1 python -m pytest tests/test_conv_pk_gpu.py tests/test_48k_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error"
for hm in 0 1; do VCVITS_PK_HALFM=$hm python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_hm$hm.json; done
