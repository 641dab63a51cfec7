"""GPU: randomized sweep of the conv family over shapes that hit every tile variant, the staging
fallbacks (wide strided spans), ragged channel / row tails, periods, groups and fused epilogues,
against torch CPU convs.  Seeds are fixed; 120 cases."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a.detach().cpu() - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def _case(rng):
    kind = rng.choice(["conv1d", "period", "convT", "grouped"], p=[0.45, 0.3, 0.15, 0.1])
    B = int(rng.integers(1, 5))
    if kind == "grouped":
        g = int(rng.choice([2, 4, 8]))
        cg, mg = int(rng.choice([1, 2, 4, 6])), int(rng.choice([1, 3, 4, 16]))
        C, M = g * cg, g * mg
    else:
        g = 1
        C = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 130, 256]))
        M = int(rng.choice([1, 2, 5, 31, 32, 33, 64, 100, 129, 260]))
    K = int(rng.choice([1, 2, 3, 4, 5, 7, 11, 16, 41])) if kind != "period" else int(rng.choice([1, 3, 5]))
    s = int(rng.choice([1, 1, 1, 2, 3, 4, 8]))
    d = int(rng.choice([1, 1, 2, 3, 5])) if s == 1 else 1
    T = int(rng.integers(max(4, (K - 1) * d + 1), 700))
    pad = int(rng.integers(0, (K * d - d) // 2 + 2))
    P = int(rng.choice([2, 3, 7, 23, 37])) if kind == "period" else 1
    if kind == "period":
        T = int(rng.integers(max(3, K), 120))
    return kind, B, C, M, T, K, s, pad, d, g, P


def test_random_conv_sweep(gpu):
    from vcvits_amd import ops
    from vcvits_amd._lib import ACT_LEAKY, ACT_NONE
    rng = np.random.default_rng(20260101)
    worst = 0.0
    for it in range(120):
        kind, B, C, M, T, K, s, pad, d, g, P = _case(rng)
        gen = torch.Generator().manual_seed(it)
        act = bool(rng.integers(0, 2))
        in_leaky = bool(rng.integers(0, 2)) and kind != "period"
        if kind == "convT":
            if K < s:
                K = s
            pad = min(pad, K - 1)
            if (T - 1) * s - 2 * pad + K < 1:
                continue
            x = torch.randn(B, C, T, generator=gen)
            w = torch.randn(C, M, K, generator=gen) / max(C * K / s, 1) ** 0.5
            b = torch.randn(M, generator=gen)
            xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
            xin = F.leaky_relu(xr, 0.1) if in_leaky else xr
            yr = F.conv_transpose1d(xin, wr, br, stride=s, padding=pad)
            xg, wg, bg = (t.clone().to(gpu).requires_grad_(True) for t in (x, w, b))
            yg = ops.conv_transpose1d(xg, wg, bg, stride=s, pad=pad, in_leaky=in_leaky, slope=0.1)
        else:
            if (T + 2 * pad - d * (K - 1) - 1) // s + 1 < 1:
                continue
            shape = (B, C, T, P) if kind == "period" else (B, C, T)
            x = torch.randn(shape, generator=gen)
            wshape = (M, C // g, K, 1) if kind == "period" else (M, C // g, K)
            w = torch.randn(wshape, generator=gen) / max(C // g * K, 1) ** 0.5
            b = torch.randn(M, generator=gen)
            xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
            xin = F.leaky_relu(xr, 0.1) if in_leaky else xr
            if kind == "period":
                yr = F.conv2d(xin, wr, br, stride=(s, 1), padding=(pad, 0), dilation=(d, 1))
            else:
                yr = F.conv1d(xin, wr, br, stride=s, padding=pad, dilation=d, groups=g)
            if act:
                yr = F.leaky_relu(yr, 0.1)
            xg, wg, bg = (t.clone().to(gpu).requires_grad_(True) for t in (x, w, b))
            yg = ops.conv1d(xg, wg, bg, stride=s, pad=pad, dil=d, groups=g, in_leaky=in_leaky,
                            out_act=ACT_LEAKY if act else ACT_NONE, slope=0.1)
        gy = torch.randn(yr.shape, generator=gen)
        yr.backward(gy)
        yg.backward(gy.to(gpu))
        tag = (it, kind, B, C, M, T, K, s, pad, d, g, P, act, in_leaky)
        for name, a, r in (("y", yg, yr.detach()), ("dx", xg.grad, xr.grad), ("dw", wg.grad, wr.grad),
                           ("db", bg.grad, br.grad)):
            assert a.shape == r.shape, (tag, name)
            e = _rel(a, r)
            worst = max(worst, e)
            assert e < 5e-5, (tag, name, e)
    print("worst rel err %.2e" % worst)
