python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_pk2.json
VCVITS_PROF_DUMP=gpurun_out/dump_voc_pk.csv python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python bench.py --workload full --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_full_pk.json
VCVITS_CONV_PK=0 python bench.py --workload full --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_full_pk0.json
