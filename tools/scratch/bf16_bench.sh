python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bf16_voc.log 2>&1
python bench.py --dtype bf16 --workload full --batch 32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2_bf16_full32.log 2>&1
python bench.py --dtype bf16 --config 48k --workload infer --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bf16_infer.log 2>&1
