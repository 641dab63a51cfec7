"""GPU: every tile variant of the split-operand conv kernel (csrc/conv_x3.hip: choose()) on the layer shapes of the bench
workload, against the variant the library's heuristic picks -- where a fixed variant beats the choice by a margin the
heuristic has something to learn.  Usage: python3 tools/x3_variant_sweep.py [--reps 10] [--only discP]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vcvits_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--only", default="")
ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
LIB = _lib.lib()
dev = torch.device("cuda:0")
B = a.batch
L = []
for ch, t in [(256, 256), (128, 2048), (64, 8192)]:
    for k in (3, 7, 11):
        L.append(("gen.res c%d k%d" % (ch, k), B, ch, ch, t, 1, k, 1, (k - 1) // 2, 1))
for p in (2, 3, 5, 7, 11, 17, 23, 37):
    h = -(-16384 // p)
    chans = [1, 32, 128, 512, 1024, 1024]
    for i in range(5):
        s = 3 if i < 4 else 1
        if i >= 2:
            L.append(("discP%d.conv%d" % (p, i), 2 * B, chans[i], chans[i + 1], h, p, 5, s, 2, 1))
        h = (h + 4 - 5) // s + 1
L.append(("discS.conv5", 2 * B, 1024, 1024, 65, 1, 5, 1, 2, 1))
VAR = ["128x256", "128x128", "256x128", "64x256", "64x128", "32x256", "64x512"]


def timeit(fn):
    try:
        for _ in range(2):
            fn()
    except RuntimeError:
        return None
    torch.cuda.synchronize()
    LIB.vcv_prof_begin(4 * a.reps + 8)
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    out = (ctypes.c_double * 15)()
    LIB.vcv_prof_end(out, 5)
    n = sum(out[3 * i] for i in range(5))
    ms = sum(out[3 * i + 1] for i in range(5))
    return ms / a.reps if n >= a.reps else None


print("%-18s %-5s | %8s | %s" % ("layer", "kind", "choice", "  ".join("%9s" % v for v in VAR)))
gain = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0]}
for name, b, c, m, t, p, k, s, pd, d in L:
    if a.only and a.only not in name:
        continue
    shape = (b, c, t) if p == 1 else (b, c, t, p)
    x = torch.randn(shape, device=dev)
    w = torch.randn(m, c, k, device=dev) * 0.05
    y = ops.conv_forward(x, w, stride=s, pad=pd, dil=d)
    dx = torch.empty_like(x)
    for kind, fn in (("fwd", lambda: ops.conv_forward(x, w, stride=s, pad=pd, dil=d, out=y)),
                     ("dgrad", lambda: ops.conv_dgrad(y, w, x.shape, stride=s, pad=pd, dil=d, out=dx))):
        LIB.vcv_conv_x3_set_variant(-1, -1, -1)
        base = timeit(fn)
        row = []
        for v in range(7):
            best = None
            for js in (1, 2):
                LIB.vcv_conv_x3_set_variant(v, js, -1)
                ops._FAMILIES.clear()
                ms = timeit(fn)
                if ms is not None and (best is None or ms < best):
                    best = ms
            row.append(best)
        LIB.vcv_conv_x3_set_variant(-1, -1, -1)
        ok = [r for r in row if r is not None]
        bestv = min(ok) if ok else None
        if base is not None and bestv is not None:
            gain[kind][0] += base
            gain[kind][1] += min(base, bestv)
        print("%-18s %-5s | %8s | %s" % (name, kind, "%.1f us" % (1e3 * base) if base else "-",
                                        "  ".join(("%7.1f%s" % (1e3 * r, "*" if r == bestv and base and r < 0.97 * base else " ")) if r else "        -" for r in row)))
for kind, (b0, b1) in gain.items():
    if b0 > 0:
        print("%s: sum of the library's choices %.3f ms, sum of the per-layer best %.3f ms (%.1f %% less)" % (kind, b0, b1, 100 * (1 - b1 / b0)))
