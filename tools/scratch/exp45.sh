python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^(FAILED)|passed|failed|^E  " | head -8
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f32.json
python bench.py --workload full --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/full.json
python bench.py --config 48k --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg5f.json
