for w in 0 1; do
echo "== t256 $w"
VCVITS_DMA_T256=$w python tools/conv_layer_bench.py --reps 10 --only "discP" 2>&1 | grep -E "conv[3-4]"
done
