"""Per-layer microbenchmark of the MFMA conv family on the layer shapes of the bench workload
(base widths, B=16): TFLOP/s of forward, data-gradient and weight-gradient launches.
Usage (GPU box): python tools/conv_layer_bench.py [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vcvits_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--only", default="")
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--split", type=int, default=6, choices=[0, 6, 9],
                help="fp32 mode: bf16 product terms of the split-operand kernel (0 = fp32-input MFMA kernels)")
a = ap.parse_args()
ops.set_compute_dtype(a.dtype)
ops.set_f32_split(a.split != 0, terms=a.split or None)
import ctypes
from vcvits_amd import _lib
LIB = _lib.lib()
dev = torch.device("cuda:0")

# name, kind, B, C, M, T(or H), P, K, stride, pad, dil, groups
L = []
B = a.batch
# generator
L.append(("gen.conv_pre", "conv", B, 256, 512, 32, 1, 7, 1, 3, 1, 1))
for i, (ci, co, t, k, s) in enumerate([(512, 256, 32, 16, 8), (256, 128, 256, 16, 8), (128, 64, 2048, 4, 4), (64, 32, 8192, 4, 2)]):
    L.append(("gen.ups%d" % i, "convT", B, ci, co, t, 1, k, s, (k - s) // 2, 1, 1))
for ch, t in [(256, 256), (128, 2048), (64, 8192), (32, 16384)]:
    for k in (3, 7, 11):
        for d in (1, 5):
            L.append(("gen.res c%d k%d d%d" % (ch, k, d), "conv", B, ch, ch, t, 1, k, 1, (k * d - d) // 2, d, 1))
L.append(("gen.conv_post", "conv", B, 32, 1, 16384, 1, 7, 1, 3, 1, 1))
# period discriminators (stacked batch 2B in the D step)
for p in (2, 37):
    h = -(-16384 // p)
    chans = [1, 32, 128, 512, 1024, 1024]
    for i in range(5):
        s = 3 if i < 4 else 1
        L.append(("discP%d.conv%d" % (p, i), "conv", 2 * B, chans[i], chans[i + 1], h, p, 5, s, 2, 1, 1))
        h = (h + 4 - 5) // s + 1
    L.append(("discP%d.post" % p, "conv", 2 * B, 1024, 1, h, p, 3, 1, 1, 1, 1))
# scale discriminator at full rate
t = 16384
for i, (ci, co, k, s, pd, g) in enumerate([(1, 16, 15, 1, 7, 1), (16, 64, 41, 4, 20, 4), (64, 256, 41, 4, 20, 16),
                                           (256, 1024, 41, 4, 20, 64), (1024, 1024, 41, 4, 20, 256), (1024, 1024, 5, 1, 2, 1)]):
    L.append(("discS.conv%d" % i, "conv", 2 * B, ci, co, t, 1, k, s, pd, 1, g))
    t = (t + 2 * pd - k) // s + 1


def timeit(fn):
    """ms per call of the GEMM kernel itself (dispatch-attached events of the library's profiler: the pack / finish
    passes around it are not included); falls back to wall time for launches the profiler does not see."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    LIB.vcv_prof_begin(4 * a.reps + 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    out = (ctypes.c_double * 15)()
    LIB.vcv_prof_end(out, 5)
    n = sum(out[3 * i] for i in range(5))
    ms = sum(out[3 * i + 1] for i in range(5))
    wall = e0.elapsed_time(e1) / a.reps
    return ms / a.reps if n >= a.reps else wall


print("%-24s %9s | %8s %8s %8s  (TFLOP/s; ms)" % ("layer", "GFLOP", "fwd", "dgrad", "wgrad"))
tot = [0.0, 0.0, 0.0, 0.0]
for name, kind, b, c, m, t, p, k, s, pd, d, g in L:
    if a.only and a.only not in name:
        continue
    shape = (b, c, t) if p == 1 else (b, c, t, p)
    x = torch.randn(shape, device=dev)
    if kind == "conv":
        w = torch.randn(m, c // g, k, device=dev) * 0.05
        y = ops.conv_forward(x, w, stride=s, pad=pd, dil=d, groups=g)
        flops = 2.0 * y.numel() * (c // g) * k
        f = lambda: ops.conv_forward(x, w, stride=s, pad=pd, dil=d, groups=g, out=y)
        dx = torch.empty_like(x)
        fd = lambda: ops.conv_dgrad(y, w, x.shape, stride=s, pad=pd, dil=d, groups=g, out=dx)
        dw = torch.zeros_like(w)
        fw = lambda: ops.conv_wgrad(y, x, w.shape, stride=s, pad=pd, dil=d, groups=g, out=dw)
    else:
        w = torch.randn(c, m, k, device=dev) * 0.05
        y = ops.convT_forward(x, w, stride=s, pad=pd)
        flops = 2.0 * x.numel() * m * k
        f = lambda: ops.convT_forward(x, w, stride=s, pad=pd, out=y)
        dx = torch.empty_like(x)
        fd = lambda: ops.convT_dgrad(y, w, x.shape, stride=s, pad=pd, out=dx)
        dw = torch.zeros_like(w)
        fw = lambda: ops.convT_wgrad(y, x, w.shape, stride=s, pad=pd, out=dw)
    ms = [timeit(f), timeit(fd), timeit(fw)]
    tf = [flops / (v * 1e-3) / 1e12 for v in ms]
    print("%-24s %9.2f | %8.1f %8.1f %8.1f   %7.3f %7.3f %7.3f" % (name, flops / 1e9, tf[0], tf[1], tf[2], ms[0], ms[1], ms[2]))
    tot[0] += flops
    for i in range(3):
        tot[i + 1] += ms[i]
print("sum GFLOP %.1f  ms fwd %.2f dgrad %.2f wgrad %.2f" % (tot[0] / 1e9, tot[1], tot[2], tot[3]))
