"""Microbenchmark of the fused attention kernels (forward + backward) at the content encoder's shapes.
python tools/attn_bench.py [--dtype bf16] [--unfused] [--reps 20]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vcvits_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32")
ap.add_argument("--unfused", action="store_true")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
ops.set_compute_dtype(a.dtype)
ops._ATTN_FUSED[0] = not a.unfused
dev = torch.device("cuda:0")
for name, B, H, dk, T in (("base B16", 16, 4, 64, 204), ("base B32", 32, 4, 64, 204), ("48k B16", 16, 4, 32, 204), ("infer 48k", 64, 4, 32, 500)):
    t = lambda *s: torch.randn(*s, device=dev)
    q, k, v, gy = (t(B, H * dk, T).requires_grad_(True) for _ in range(4))
    ek, ev = t(1, 9, dk).requires_grad_(True), t(1, 9, dk).requires_grad_(True)
    mask = torch.ones(B, T, device=dev)

    def fwd():
        with torch.no_grad():
            return ops.rel_attention(q, k, v, ek, ev, mask, H, 4, want_attn=False)

    def fb():
        o, _ = ops.rel_attention(q, k, v, ek, ev, mask, H, 4, 0.1, training=True, want_attn=False)
        o.backward(gy)

    import ctypes
    from vcvits_amd import _lib
    LIB = _lib.lib()
    res = []
    for f, mult in ((fwd, 1.0), (fb, 3.0)):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        LIB.vcv_prof_begin(8 * a.reps + 8)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        out = (ctypes.c_double * 15)()
        LIB.vcv_prof_end(out, 5)
        kern = (out[1] + out[13]) / a.reps * 1e-3 if (out[0] + out[12]) > 0 else float("nan")  # GPU time of the GEMM-shaped launches
        flops = mult * 4.0 * B * H * T * T * dk
        res.append((dt * 1e6, flops / dt / 1e12, kern * 1e6, flops / kern / 1e12))
    print("%-10s B=%d H=%d dk=%d T=%d | fwd %7.1f us wall, %7.1f us in the kernels: %6.1f TFLOP/s | fwd+bwd %7.1f us wall, %7.1f us in the kernels: %6.1f TFLOP/s" % (
        name, B, H, dk, T, res[0][0], res[0][2], res[0][3], res[1][0], res[1][2], res[1][3]))
