python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED|PASSED)|passed|failed|Error" | head -20
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_wg.json
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --workload full 2>/dev/null | tail -1 > gpurun_out/r2_full_wg.json
