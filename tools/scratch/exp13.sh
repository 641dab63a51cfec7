for pk in 0 1; do VCVITS_CONV_PK=$pk python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_pk$pk.json; done
VCVITS_CONV_PK=1 VCVITS_PROF_DUMP=gpurun_out/dump_voc_pk.csv python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
