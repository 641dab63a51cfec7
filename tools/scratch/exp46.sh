for e in "XX=1" "PK_NO512=1"; do echo "CFG $e"; env $e python tools/conv_layer_bench.py --reps 10 --only "gen.res c32" 2>/dev/null | cut -c1-100; done
