for e in "XX=1" "PK_WS_MORE=1" "PK_WS_MORE=4"; do echo "CFG $e"; env $e python tools/conv_layer_bench.py --reps 10 --only discP 2>/dev/null | cut -c1-100; done
for e in "XX=1" "PK_WS_MORE=2"; do echo "CFG bf16 $e"; env $e python tools/conv_layer_bench.py --reps 10 --only discP --dtype bf16 2>/dev/null | cut -c1-100; done
