python -m pytest tests -m gpu -q -x 2>&1 | grep -E "^E  +(Assertion|assert|Runtime)|^(FAILED)|passed|failed" | head -8
for e in "VCVITS_PK_NO_VEC=1" "XX=1"; do env $e python bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bf16_$e.json; env $e python bench.py --dtype bf16 --config 48k --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg5_$e.json; done
python bench.py --config 48k --workload infer --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg5f.json
python bench.py --workload full --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/full.json
