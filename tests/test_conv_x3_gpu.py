"""GPU: the split-operand fp32 conv kernel (vcv_conv_x3_*: every fp32 operand as three exact bf16 terms, nine -- or six --
bf16 MFMA products per fp32 product, fp32 accumulate) against float64 torch CPU convolutions, next to the fp32-input MFMA
kernel (vcv_conv_pk_*, an fmaf chain) on the same inputs: forward, data gradient (stride-1 flipped, phased strided,
ConvTranspose) on layer shapes of both configs.  The claim under test: the split kernel is fp32 arithmetic -- its error
against the float64 result is of the size of the fmaf chain's own, 2e-5 max-norm as for every other fp32 kernel here."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import record_stats
from test_bf16_gpu import CASES

pytestmark = pytest.mark.gpu

EXTRA = [
    ("conv", 2, 512, 512, 300, 7, 1, 3, 1, 1, False),     # generator conv_pre-like, ragged tile edge
    ("conv", 16, 1024, 1024, 64, 5, 1, 2, 1, 1, False),   # DiscriminatorS conv5 (batch folded into columns, split reduction)
    ("conv", 2, 48, 40, 700, 9, 1, 4, 1, 1, True),        # channel tails on both sides, 64-row tile with a ragged m-tile
    ("conv", 1, 16, 32, 5000, 16, 1, 8, 1, 1, False),     # one channel group, 16 taps
    ("conv", 3, 96, 96, 130, 1, 1, 0, 1, 1, False),       # 1x1
]


def rel64(a, b):
    return (a.detach().cpu().double() - b).abs().max().item() / (b.abs().max().item() + 1e-300)


@pytest.mark.parametrize("case", [c for c in CASES if c[2] >= 16] + EXTRA,
                         ids=lambda c: "%s-C%d-M%d-T%d-K%d-s%d-d%d-P%d" % (c[0], c[2], c[3], c[4], c[5], c[6], c[8], c[9]))
def test_x3_kernel_is_fp32_arithmetic(gpu, case):
    from vcvits_amd import ops
    kind, B, C, M, T, K, s, pad, d, P, in_leaky = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    if kind == "convT":
        x, w, b = t(B, C, T), t(C, M, K) * (C * K / s) ** -0.5, t(M) * 0.1
    elif kind == "period":
        x, w, b = t(B, C, T, P), t(M, C, K, 1) * (C * K) ** -0.5, t(M) * 0.1
    else:
        x, w, b = t(B, C, T), t(M, C, K) * (C * K) ** -0.5, t(M) * 0.1
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    xin = F.leaky_relu(xr, 0.1) if in_leaky else xr
    if kind == "convT":
        yr = F.conv_transpose1d(xin, wr, b.double(), stride=s, padding=pad)
    elif kind == "period":
        yr = F.conv2d(xin, wr, b.double(), stride=(s, 1), padding=(pad, 0))
    else:
        yr = F.conv1d(xin, wr, b.double(), stride=s, padding=pad, dilation=d)
    gy = t(*yr.shape)
    yr.backward(gy.double())
    errs = {}
    try:
        for mode in ("x3-9", "x3-6", "pk"):
            ops.set_f32_split(mode != "pk", terms=9 if mode != "x3-6" else 6, all_shapes=True)
            before = dict(ops.LAUNCH_COUNTS)
            xg, wg, bg = (v.to(gpu).requires_grad_(True) for v in (x, w, b))
            if kind == "convT":
                yg = ops.conv_transpose1d(xg, wg, bg, stride=s, pad=pad, in_leaky=in_leaky, slope=0.1)
            else:
                yg = ops.conv1d(xg, wg, bg, stride=s, pad=pad, dil=d, in_leaky=in_leaky, slope=0.1)
            yg.backward(gy.to(gpu))
            used = ops.LAUNCH_COUNTS["x3"] - before["x3"]
            if mode == "pk":
                assert used == 0
            else:
                assert used >= 1, "the split kernel took none of this case's launches"
            errs[mode] = (rel64(yg, yr.detach()), rel64(xg.grad, xr.grad))
    finally:
        ops.set_f32_split(True, terms=6, all_shapes=False)
    record_stats("x3", "%s-C%d-M%d-T%d-K%d-s%d-P%d" % (kind, C, M, T, K, s, P),
                 **{"%s_%s" % (m.replace("-", "_"), n): v for m, e in errs.items() for n, v in zip(("y", "dx"), e)})
    for mode, (ey, ex) in errs.items():
        assert ey < 2e-5 and ex < 2e-5, (mode, ey, ex)
    # same error class as the fmaf chain (both are dominated by the fp32 accumulation order)
    for i in range(2):
        assert errs["x3-9"][i] <= 3 * errs["pk"][i] + 2e-7, errs
        assert errs["x3-6"][i] <= 4 * errs["pk"][i] + 4e-7, errs


def test_split_is_exact(gpu):
    """x = x0 + x1 + x2 exactly: a 1x1 convolution of a one-hot weight row with NTERM = 9 reproduces the input bits."""
    from vcvits_amd import ops
    rng = np.random.default_rng(5)
    x = torch.from_numpy((rng.standard_normal((2, 32, 512)) * np.exp(rng.uniform(-20, 20, (2, 32, 512)))).astype(np.float32))
    w = torch.zeros(32, 32, 1)
    for m in range(32):
        w[m, (m * 7) % 32, 0] = 1.0
    ops.set_f32_split(True, terms=9, all_shapes=True)
    try:
        before = ops.LAUNCH_COUNTS["x3"]
        y = ops.conv_forward(x.to(gpu), w.to(gpu))
        assert ops.LAUNCH_COUNTS["x3"] == before + 1
    finally:
        ops.set_f32_split(True, terms=6, all_shapes=False)
    assert torch.equal(y.cpu(), x[:, [(m * 7) % 32 for m in range(32)], :])


# (ConvTranspose weight gradients with stride > 3 stay on the fp32 kernels, as in bf16 mode)
WG_CASES = [c for c in CASES if c[2] >= 16 and not (c[0] == "convT" and c[6] > 3)] + EXTRA[:3]


@pytest.mark.parametrize("case", WG_CASES,
                         ids=lambda c: "%s-C%d-M%d-T%d-K%d-s%d-d%d-P%d" % (c[0], c[2], c[3], c[4], c[5], c[6], c[8], c[9]))
def test_x3_weight_gradient_is_fp32_arithmetic(gpu, case):
    """Weight (and bias) gradient on the split-operand kernel (vcv_wgrad_x3) vs float64, next to the fp32-input MFMA
    kernel; bit-reproducible between two identical launches (slabs added in a fixed order)."""
    from vcvits_amd import ops
    kind, B, C, M, T, K, s, pad, d, P, in_leaky = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 1)
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    if kind == "convT":
        x, w = t(B, C, T), t(C, M, K) * (C * K / s) ** -0.5
    elif kind == "period":
        x, w = t(B, C, T, P), t(M, C, K, 1) * (C * K) ** -0.5
    else:
        x, w = t(B, C, T), t(M, C, K) * (C * K) ** -0.5
    b = t(M) * 0.1
    xr, wr, br = x.double(), w.double().requires_grad_(True), b.double().requires_grad_(True)
    xin = F.leaky_relu(xr, 0.1) if in_leaky else xr
    if kind == "convT":
        yr = F.conv_transpose1d(xin, wr, br, stride=s, padding=pad)
    elif kind == "period":
        yr = F.conv2d(xin, wr, br, stride=(s, 1), padding=(pad, 0))
    else:
        yr = F.conv1d(xin, wr, br, stride=s, padding=pad, dilation=d)
    gy = t(*yr.shape)
    yr.backward(gy.double())
    errs, grads = {}, {}
    try:
        for mode in ("x3-9", "x3-9b", "x3-6", "pk"):
            ops.set_f32_split(mode != "pk", terms=6 if mode == "x3-6" else 9, wgrad=True, all_shapes=True)
            before = dict(ops.LAUNCH_COUNTS)
            xg, wg, bg = x.to(gpu), w.to(gpu).requires_grad_(True), b.to(gpu).requires_grad_(True)
            if kind == "convT":
                yg = ops.conv_transpose1d(xg, wg, bg, stride=s, pad=pad, in_leaky=in_leaky, slope=0.1)
            else:
                yg = ops.conv1d(xg, wg, bg, stride=s, pad=pad, dil=d, in_leaky=in_leaky, slope=0.1)
            yg.backward(gy.to(gpu))
            used = ops.LAUNCH_COUNTS["wgrad_x3"] - before["wgrad_x3"]
            if mode == "pk":
                assert used == 0
            elif used == 0:
                ops.set_f32_split(True, terms=6, wgrad=True, all_shapes=False)
                pytest.skip("no tile of the split weight-gradient kernel fits this shape (the fp32 kernel keeps it)")
            errs[mode] = (rel64(wg.grad, wr.grad), rel64(bg.grad, br.grad))
            grads[mode] = wg.grad.detach().clone()
    finally:
        ops.set_f32_split(True, terms=6, wgrad=True, all_shapes=False)
    record_stats("x3wgrad", "%s-C%d-M%d-T%d-K%d-s%d-P%d" % (kind, C, M, T, K, s, P),
                 **{"%s_%s" % (m.replace("-", "_"), n): v for m, e in errs.items() for n, v in zip(("dw", "db"), e)})
    assert torch.equal(grads["x3-9"], grads["x3-9b"]), "two identical launches differ"
    for mode, (ew, eb) in errs.items():
        assert ew < 3e-5 and eb < 3e-5, (mode, ew, eb)
    assert errs["x3-9"][0] <= 3 * errs["pk"][0] + 3e-7 and errs["x3-6"][0] <= 4 * errs["pk"][0] + 6e-7, errs
