"""Per-launch times of the one-output-channel / one-input-channel convolutions at the training step's shapes
(discriminator heads and first layers at every period / pooled scale, the generator's conv_post).
python tools/thin_bench.py [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vcvits_amd import _lib, ops
from vcvits_amd._lib import ACT_NONE, ACT_TANH, TF_LEAKY, TF_NONE

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
LIB = _lib.lib()


def timed(f):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3  # us per launch (back-to-back launches on the stream)


def period_rows(p, seg=8192):
    h = -(-seg // p)
    for _ in range(4):
        h = (h + 4 - 5) // 3 + 1
    return h


cases = []
for p in (2, 3, 5, 7, 11, 17, 23, 37):
    cases.append(("discP%d.post" % p, 64, 1024, 1, period_rows(p), p, 3, 1, 1, TF_NONE, ACT_NONE))
for i in range(5):
    cases.append(("discS/%d.post" % (1 << i), 64, 1024, 1, 8192 // (1 << i) // 256, 1, 3, 1, 1, TF_NONE, ACT_NONE))
cases.append(("gen.conv_post", 32, 32, 1, 8192, 1, 7, 1, 3, TF_LEAKY, ACT_TANH))
for p in (2, 37):
    cases.append(("discP%d.conv0" % p, 64, 1, 32, -(-8192 // p), p, 5, 3, 2, TF_NONE, ACT_NONE))
for i in (0, 2):
    cases.append(("discS/%d.conv0" % (1 << i), 64, 1, 16, 8192 >> i, 1, 15, 1, 7, TF_NONE, ACT_NONE))

tot = [0.0, 0.0, 0.0]
print("%-16s %5s %5s %3s %6s %3s | %8s %8s %8s %8s  (us per launch)" % ("layer", "C", "M", "K", "rows", "P", "fwd", "wgrad", "dgrad", "MB"))
for name, B, C, M, H, P, K, s, pad, in_tf, act in cases:
    x = torch.randn(B, C, H, P, device=dev) if P > 1 else torch.randn(B, C, H, device=dev)
    w = torch.randn(M, C, K, device=dev)
    bias = torch.randn(M, device=dev)
    y = ops.conv_forward(x, w, bias, stride=s, pad=pad, in_tf=in_tf, out_act=act)
    gy = torch.randn_like(y)
    tf = timed(lambda: ops.conv_forward(x, w, bias, stride=s, pad=pad, in_tf=in_tf, out_act=act))
    tw = timed(lambda: ops.conv_wgrad(gy, x, (M, C, K), stride=s, pad=pad))
    td = timed(lambda: ops.conv_dgrad(gy, w, x.shape, stride=s, pad=pad))
    tot[0] += tf
    tot[1] += tw
    tot[2] += td
    print("%-16s %5d %5d %3d %6d %3d | %8.1f %8.1f %8.1f %8.1f" % (name, C, M, K, H, P, tf, tw, td, (x.numel() + y.numel()) * 4 / 1e6))
print("sum fwd %.1f us, wgrad %.1f us, dgrad %.1f us" % tuple(tot))
