python -m pytest tests/test_conv_gpu.py tests/test_elementwise_gpu.py tests/test_modules_gpu.py tests/test_conv_pk_gpu.py -m gpu -q 2>&1 | grep -E "^(FAILED)|passed|failed" | head -4
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/now/trace -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/now.trace.log 2>&1
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f32.json
