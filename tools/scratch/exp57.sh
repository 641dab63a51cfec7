python -m pytest tests/test_elementwise_gpu.py tests/test_conv_gpu.py tests/test_modules_gpu.py tests/test_training_step_gpu.py -m gpu -q 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED)|passed|failed" | head -8
for i in 1 2; do python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['share_of_step_time'])"; done
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/now3/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/now3.trace.log 2>&1
