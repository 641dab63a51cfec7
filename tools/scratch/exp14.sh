for cap in 0 1 2; do VCVITS_PK_CAP=$cap VCVITS_CONV_PK=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_pkcap$cap.json; done
