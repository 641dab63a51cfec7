python -m pytest tests/test_bf16_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error"
for cap in 156 78; do
echo "== cap $cap"
VCVITS_BF16_LDSCAP=$cap python tools/conv_layer_bench.py --dtype bf16 --reps 10 --only discP 2>&1 | grep -E "conv[1-4]"
VCVITS_BF16_LDSCAP=$cap python tools/conv_layer_bench.py --dtype bf16 --reps 10 --only "gen.res c" 2>&1 | grep -E "d1 "
done
