python -m pytest tests/test_conv_pk_gpu.py tests/test_conv_gpu.py tests/test_conv_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED)|passed|failed" | head -5
for i in 1 2; do python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['share_of_step_time'])"; done
