WG_VERBOSE=1 python tools/conv_layer_bench.py --reps 1 --only "k7 d1" 2>&1 | grep "wgrad plan" | sort -u | head -60
for z in 1 2 4 8 16 32 64 128; do echo "CFG Z$z"; WG_TILE=1 WG_Z=$z python tools/conv_layer_bench.py --reps 8 --only "k7 d1" 2>/dev/null | cut -c1-100; WG_TILE=1 WG_Z=$z python tools/conv_layer_bench.py --reps 8 --only "discP2.conv2" 2>/dev/null | grep discP | cut -c1-100; done
for z in 16 32 64 128 256 512; do echo "CFG t6Z$z"; WG_TILE=6 WG_Z=$z python tools/conv_layer_bench.py --reps 8 --only "c32 k7 d1" 2>/dev/null | cut -c1-100; done
