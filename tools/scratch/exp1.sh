for cap in 156 78; do for n in 0 1; do
echo "== cap $cap no224 $n"
VCVITS_BF16_LDSCAP=$cap VCVITS_BF16_NO224=$n python tools/conv_layer_bench.py --dtype bf16 --reps 10 --only discP 2>&1 | grep -E "conv[1-4]"
VCVITS_BF16_LDSCAP=$cap VCVITS_BF16_NO224=$n python tools/conv_layer_bench.py --dtype bf16 --reps 10 --only "gen.res c" 2>&1 | grep -E "d1 "
done; done
