for e in "VCVITS_PK_NO_X4=1" "XX=1"; do echo "CFG $e"; env $e python tools/conv_layer_bench.py --reps 10 --dtype bf16 2>/dev/null | cut -c1-100; done
