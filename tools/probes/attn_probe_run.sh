#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_attention_fused_gpu.py tests/test_modules_gpu.py -x -q 2>&1 | tail -3
echo "== new f32"; timeout 300 python tools/attn_bench.py 2>&1 | tail -4
echo "== new bf16"; timeout 300 python tools/attn_bench.py --dtype bf16 2>&1 | tail -4
export VCVITS_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libvcvits_stamps.so; python tools/probes/attn_stamps.py f32 2>/dev/null | head -10; python tools/probes/attn_stamps.py bf16 2>/dev/null | head -10
