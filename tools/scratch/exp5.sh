for w in 0 1 3; do
echo "== waves16 $w"
VCVITS_DMA_WAVES16=$w python tools/conv_layer_bench.py --reps 10 --only "gen.res c" 2>&1 | grep -E "d1 "
VCVITS_DMA_WAVES16=$w python tools/conv_layer_bench.py --reps 10 --only "discP2" 2>&1 | grep -E "conv[1-4]"
done
