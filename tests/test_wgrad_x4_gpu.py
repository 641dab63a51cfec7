"""GPU: the warp-specialised fp32 weight-gradient kernel (wgrad_dma.hip: producer waves, 16-byte fragments, the
per-position offset table of strided launches) against torch CPU autograd on shapes chosen to hit its edges:
sequence lengths that are not multiples of 4 / 16 / 64 (partial last stage, element masks at the row ends, unaligned
16-byte loads), periods with and without stride, strides up to 8 with fewer taps than the stride, dilation,
transposed-conv roles, channel counts that are not multiples of the 32-row MFMA tile, batch 1, the bias row sums, the
input leaky-ReLU on either operand and the deterministic slab combine.  Exact fp32: 2e-5 max-norm (summation order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fp32_weight_gradient_kernel():
    """These tests are about wgrad_dma.hip: keep the split-operand weight gradient (which takes some of these shapes in the
    default fp32 mode: tests/test_conv_x3_gpu.py) out of the way."""
    from vcvits_amd import ops
    ops._USE_X3_WGRAD[0] = False
    yield
    ops._USE_X3_WGRAD[0] = True

# kind, B, C, M, T (rows), P, K, stride, pad, dil, in_leaky
CASES = [
    ("conv", 1, 33, 40, 67, 1, 3, 1, 1, 1, False),        # one sequence, odd channel counts, U = 67
    ("conv", 3, 64, 96, 130, 1, 7, 1, 9, 3, True),        # dilation 3, U = 130, input leaky
    ("conv", 2, 128, 128, 1021, 1, 11, 1, 25, 5, False),  # U = 1021 (prime): every row start misaligned
    ("conv", 2, 200, 72, 64, 1, 5, 1, 2, 1, False),       # exactly one stage
    ("conv", 2, 48, 260, 65, 1, 5, 1, 2, 1, False),       # one position in the second stage
    ("conv", 2, 64, 64, 81, 1, 4, 2, 1, 1, False),        # stride 2, even taps
    ("conv", 2, 40, 136, 203, 1, 3, 4, 1, 1, False),      # stride 4 > K - 1: a residue without taps
    ("conv", 2, 32, 128, 530, 1, 3, 8, 0, 1, False),      # stride 8, K 3
    ("conv", 2, 32, 128, 529, 1, 16, 8, 4, 1, True),      # K 16, stride 8 (the transposed convs' roles, as a conv)
    ("period", 2, 32, 128, 70, 3, 5, 3, 2, 1, False),     # U = 72
    ("period", 2, 64, 256, 41, 5, 5, 3, 2, 1, True),
    ("period", 1, 128, 160, 23, 13, 5, 3, 2, 1, False),   # P 13: spans of several hundred floats
    ("period", 2, 96, 128, 9, 37, 5, 3, 2, 1, False),     # P 37, three output rows
    ("period", 2, 256, 256, 6, 37, 5, 1, 2, 1, False),    # P 37 stride 1: U = 222
    ("period", 2, 128, 192, 12, 17, 5, 1, 2, 1, True),
    ("period", 3, 64, 64, 35, 2, 5, 1, 2, 1, False),
    ("convT", 2, 96, 48, 33, 1, 16, 8, 4, 1, True),
    ("convT", 2, 64, 64, 100, 1, 4, 2, 1, 1, False),
    ("convT", 1, 40, 36, 70, 1, 7, 3, 2, 1, False),
]


def _ids(c):
    return "%s-B%d-C%d-M%d-T%d-P%d-K%d-s%d-p%d-d%d-%s" % (c[:10] + ("leaky" if c[10] else "lin",))


def rel(a, b):
    return float((a.detach().cpu() - b.detach().cpu()).abs().max() / b.detach().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("det", [False, True], ids=["atomics", "slabs"])
@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_weight_gradient_matches_torch(gpu, case, det):
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
    ops.set_deterministic(det)
    try:
        before = ops.LAUNCH_COUNTS["wgrad"]
        xg, wg, bg = (v.to(gpu).requires_grad_(True) for v in (x, w, b))
        if kind == "convT":
            yg = ops.conv_transpose1d(xg, wg, bg, stride=s, pad=pad, in_leaky=in_leaky, slope=0.1)
        else:
            yg = ops.conv1d(xg, wg, bg, stride=s, pad=pad, dil=d, in_leaky=in_leaky, slope=0.1)
        yg.backward(gy.to(gpu))
        assert ops.LAUNCH_COUNTS["wgrad"] > before
    finally:
        ops.set_deterministic(False)
    assert rel(wg.grad, wr.grad) < 2e-5, "dw"
    assert rel(bg.grad, br.grad) < 2e-5, "db"
    assert rel(xg.grad, xr.grad) < 2e-5, "dx"


def test_weight_gradient_accumulates_and_scales(gpu):
    """`out` holds a value to accumulate onto, alpha scales the new term, dbias is added onto its buffer."""
    from vcvits_amd import ops
    rng = np.random.default_rng(5)
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    B, C, M, T, K = 2, 72, 136, 333, 5
    x, dy, w0, b0 = t(B, C, T), t(B, M, T), t(M, C, K), t(M)
    xr = x.clone()
    wr = torch.zeros(M, C, K, requires_grad=True)
    F.conv1d(xr, wr, None, padding=2).backward(dy)
    out, db = w0.clone().to(gpu), b0.clone().to(gpu)
    ops.conv_wgrad(dy.to(gpu), x.to(gpu), (M, C, K), stride=1, pad=2, out=out, alpha=0.5, dbias=db)
    assert rel(out, w0 + 0.5 * wr.grad) < 2e-5
    assert rel(db, b0 + dy.sum(dim=(0, 2))) < 2e-5
