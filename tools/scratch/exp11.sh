python -m pytest tests/test_bf16_gpu.py -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_bf16d/trace -o t -- python3 bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_bf16_voc.json
