for e in "WG_DBG=0" "WG_DBG=4" "WG_DBG=7"; do echo "CFG $e"; env $e python tools/conv_layer_bench.py --reps 10 --only discP 2>/dev/null | cut -c1-100; done
