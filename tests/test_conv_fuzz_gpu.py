"""GPU: randomised shapes through the fast conv family (packed-operand kernel with 16-byte loads / producer waves,
warp-specialised weight gradient) against torch CPU autograd.  The fixed cases of the other files pin the layer shapes
of the two configs; this sweep draws channel counts, lengths, periods, kernel sizes, strides, paddings and dilations
at random so that tile edges, partial stages, misaligned rows and the zero padding at both sequence ends are hit in
combinations nobody wrote down (the immediate-offset folding bug of round 2 lived in exactly such a corner).
Exact fp32: 2e-5 max-norm."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _draw(rng, i):
    kind = ("conv", "period", "convT")[i % 3]
    B = int(rng.integers(1, 4))
    C = int(rng.choice([32, 40, 64, 96, 128, 200, 256]))
    M = int(rng.choice([32, 48, 64, 128, 160, 256, 320]))
    in_leaky = bool(rng.integers(0, 2))
    if kind == "conv":
        K = int(rng.choice([3, 5, 7, 11]))
        d = int(rng.choice([1, 1, 3, 5]))
        s = 1 if d > 1 else int(rng.choice([1, 1, 2, 3]))
        pad = int(rng.integers(0, (K - 1) * d + 1))
        T = int(rng.integers(70, 700))
        P = 1
    elif kind == "period":
        K, d = 5, 1
        s = int(rng.choice([1, 3]))
        pad = 2
        P = int(rng.choice([2, 3, 5, 7, 11, 13, 17, 23, 37]))
        T = int(rng.integers(max(6, 80 // P), max(12, 700 // P)))
    else:
        s = int(rng.choice([2, 4, 8]))
        K = int(rng.choice([s, 2 * s]))
        d = 1
        pad = (K - s) // 2
        T = int(rng.integers(20, 200))
        P = 1
    return kind, B, C, M, T, P, K, s, pad, d, in_leaky


# VCVITS_FUZZ_N widens the sweep (400 shapes were run once at the end of round 2; the default keeps the suite short)
CASES = [_draw(np.random.default_rng(1000 + i), i) for i in range(int(os.environ.get("VCVITS_FUZZ_N", "36")))]


def rel(a, b):
    return float((a.detach().cpu() - b.detach().cpu()).abs().max() / b.detach().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-B%d-C%d-M%d-T%d-P%d-K%d-s%d-p%d-d%d-%s" % (c[:10] + ("leaky" if c[10] else "lin",)))
def test_random_shape_matches_torch(gpu, case):
    from vcvits_amd import ops
    kind, B, C, M, T, P, K, s, pad, d, in_leaky = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    if kind == "convT":
        x, w, b = t(B, C, T), t(C, M, K) * (C * K / s) ** -0.5, t(M) * 0.1
    elif kind == "period":
        x, w, b = t(B, C, T, P), t(M, C, K, 1) * (C * K) ** -0.5, t(M) * 0.1
    else:
        x, w, b = t(B, C, T), t(M, C, K) * (C * K) ** -0.5, t(M) * 0.1
    xr, wr, br = (v.clone().requires_grad_(True) for v in (x, w, b))
    xin = F.leaky_relu(xr, 0.1) if in_leaky else xr
    if kind == "convT":
        yr = F.conv_transpose1d(xin, wr, br, stride=s, padding=pad)
    elif kind == "period":
        yr = F.conv2d(xin, wr, br, stride=(s, 1), padding=(pad, 0))
    else:
        yr = F.conv1d(xin, wr, br, stride=s, padding=pad, dilation=d)
    gy = t(*yr.shape)
    yr.backward(gy)
    xg, wg, bg = (v.to(gpu).requires_grad_(True) for v in (x, w, b))
    if kind == "convT":
        yg = ops.conv_transpose1d(xg, wg, bg, stride=s, pad=pad, in_leaky=in_leaky, slope=0.1)
    else:
        yg = ops.conv1d(xg, wg, bg, stride=s, pad=pad, dil=d, in_leaky=in_leaky, slope=0.1)
    yg.backward(gy.to(gpu))
    assert rel(yg, yr) < 2e-5, "y"
    assert rel(xg.grad, xr.grad) < 2e-5, "dx"
    assert rel(wg.grad, wr.grad) < 3e-5, "dw"
    assert rel(bg.grad, br.grad) < 2e-5, "db"
