python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^(FAILED)|passed|failed" | head -8
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/f32.json
python bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bf16.json
python bench.py --dtype bf16 --config 48k --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg5.json
python bench.py --dtype bf16 --workload full --batch 32 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg3.json
