for w in "0 0" "1 0" "0 1"; do set -- $w
echo "== wide32 $1 wide64 $2"
VCVITS_BF16_WIDE32=$1 VCVITS_BF16_WIDE64=$2 python tools/conv_layer_bench.py --dtype bf16 --reps 10 --only "gen.res c" 2>&1 | grep -E "c(32|64) .*d1 "
done
VCVITS_BF16_WIDE32=1 VCVITS_BF16_WIDE64=1 python -m pytest tests/test_bf16_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error"
