python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED)|passed|failed" | head -8
python bench.py --workload full --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/full.json
python bench.py --dtype bf16 --workload full --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg3.json
