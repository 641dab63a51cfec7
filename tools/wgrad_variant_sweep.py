"""GPU: tile candidates x reduction splits of the bf16 / split-operand weight-gradient kernel (csrc/wgrad_bf16.hip: pick(),
launch()) on the layer shapes of the bench workload, against the library's own choice.
Usage: python3 tools/wgrad_variant_sweep.py [--dtype bf16|f32] [--reps 8] [--only discP]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vcvits_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--only", default="")
ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"])
ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
ops.set_compute_dtype(a.dtype)
LIB = _lib.lib()
dev = torch.device("cuda:0")
B = a.batch
L = []
for ch, t in [(256, 256), (128, 2048), (64, 8192)]:
    for k in (3, 7, 11):
        L.append(("gen.res c%d k%d" % (ch, k), B, ch, ch, t, 1, k, 1, (k - 1) // 2))
for p in (2, 5, 11, 23, 37):
    h = -(-16384 // p)
    chans = [1, 32, 128, 512, 1024, 1024]
    for i in range(5):
        s = 3 if i < 4 else 1
        if i >= 2:
            L.append(("discP%d.conv%d" % (p, i), 2 * B, chans[i], chans[i + 1], h, p, 5, s, 2))
        h = (h + 4 - 5) // s + 1
L.append(("discS.conv5", 2 * B, 1024, 1024, 65, 1, 5, 1, 2))
CAND = ["128x64", "128x32", "64x64", "64x32", "32x64", "32x32"]
ZS = [-1, 1, 2, 4, 8, 16, 32]


def timeit(fn):
    try:
        for _ in range(2):
            fn()
    except RuntimeError:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps  # (wall per call: the kernel AND its finishing pass)


print("# wall time per weight-gradient call (kernel + finishing pass), us; * = beats the library's choice by > 3 %")
print("%-18s | %8s | %s" % ("layer", "choice", "  ".join("%14s" % c for c in CAND)))
tot = [0.0, 0.0]
for name, b, c, m, t, p, k, s, pd in L:
    if a.only and a.only not in name:
        continue
    shape = (b, c, t) if p == 1 else (b, c, t, p)
    x = torch.randn(shape, device=dev)
    w = torch.randn(m, c, k, device=dev) * 0.05
    y = ops.conv_forward(x, w, stride=s, pad=pd)
    dw = torch.zeros_like(w)
    fn = lambda: ops.conv_wgrad(y, x, w.shape, stride=s, pad=pd, out=dw)
    LIB.vcv_wgrad_bf16_set_force(-1, -1)
    base = timeit(fn)
    row = []
    for ci in range(6):
        best, bz = None, None
        for z in ZS:
            LIB.vcv_wgrad_bf16_set_force(ci, z)
            ms = timeit(fn)
            if ms is not None and (best is None or ms < best):
                best, bz = ms, z
        row.append((best, bz))
    LIB.vcv_wgrad_bf16_set_force(-1, -1)
    ok = [r[0] for r in row if r[0] is not None]
    bestv = min(ok) if ok else None
    if base and bestv:
        tot[0] += base
        tot[1] += min(base, bestv)
    print("%-18s | %8s | %s" % (name, "%.1f" % (1e3 * base) if base else "-",
                              "  ".join(("%7.1f%s z%-4s" % (1e3 * r, "*" if base and r < 0.97 * base and r == bestv else " ", z)) if r else "             -" for r, z in row)))
if tot[0] > 0:
    print("sum of the library's choices %.3f ms, sum of the per-layer best %.3f ms (%.1f %% less)" % (tot[0], tot[1], 100 * (1 - tot[1] / tot[0])))
