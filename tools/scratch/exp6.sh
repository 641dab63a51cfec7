for w in 0 1; do
echo "== deep $w"
VCVITS_DMA_DEEP=$w python tools/conv_layer_bench.py --reps 10 --only "discP" 2>&1 | grep -E "conv[1-4]"
VCVITS_DMA_DEEP=$w python tools/conv_layer_bench.py --reps 10 --only "gen.res c256" 2>&1 | grep -E "d1 "
done
VCVITS_DMA_DEEP=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_deep.json
