for e in "PK_WS=0" "PK_WS=7" "PK_WS=15"; do echo "CFG $e"; env $e python tools/conv_layer_bench.py --reps 10 2>/dev/null | cut -c1-100; done
