for t in 0 1 2 3 4 5 6; do echo "CFG tile$t"; WG_TILE=$t python tools/conv_layer_bench.py --reps 8 2>/dev/null | cut -c1-100; done
