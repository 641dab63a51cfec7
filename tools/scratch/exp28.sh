WG_VERBOSE=1 python tools/conv_layer_bench.py --reps 10 2>gpurun_out/plans.txt > gpurun_out/layers_wg3.txt
python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_f32_wg.json
