"""GPU: the fp32 packed-operand conv kernel (vcv_conv_pk_*: channel-innermost 16-byte LDS groups, four
v_mfma_f32_32x32x2_f32 per fragment pair) against torch CPU convs AND against the LDS-DMA kernel it replaces for these
shapes -- forward and data gradient, on the layer shapes of both configs.  Exact fp32: 2e-5 max-norm (summation order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_bf16_gpu import CASES, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [c for c in CASES if c[2] >= 64],
                         ids=lambda c: "%s-C%d-M%d-T%d-K%d-s%d-d%d-P%d" % (c[0], c[2], c[3], c[4], c[5], c[6], c[8], c[9]))
def test_pk_kernel_matches_torch_and_dma(gpu, case):
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
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    xin = F.leaky_relu(xr, 0.1) if in_leaky else xr
    if kind == "convT":
        yr = F.conv_transpose1d(xin, wr, b, stride=s, padding=pad)
    elif kind == "period":
        yr = F.conv2d(xin, wr, b, stride=(s, 1), padding=(pad, 0))
    else:
        yr = F.conv1d(xin, wr, b, stride=s, padding=pad, dilation=d)
    gy = t(*yr.shape)
    yr.backward(gy)
    outs = {}
    for use_pk in (True, False):
        ops._USE_PK[0] = use_pk
        ops._USE_X3[0] = False  # (the split-operand kernel would take these launches first: tests/test_conv_x3_gpu.py)
        try:
            before = ops.LAUNCH_COUNTS["pk"]
            xg, wg, bg = (v.to(gpu).requires_grad_(True) for v in (x, w, b))
            if kind == "convT":
                yg = ops.conv_transpose1d(xg, wg, bg, stride=s, pad=pad, in_leaky=in_leaky, slope=0.1)
            else:
                yg = ops.conv1d(xg, wg, bg, stride=s, pad=pad, dil=d, in_leaky=in_leaky, slope=0.1)
            yg.backward(gy.to(gpu))
            used = ops.LAUNCH_COUNTS["pk"] - before
        finally:
            ops._USE_PK[0] = True
            ops._USE_X3[0] = True
        if use_pk and not (K <= 3 and C >= 128):
            assert used >= 1, "the packed kernel did not take this launch"
        if not use_pk:
            assert used == 0
        assert rel(yg, yr.detach()) < 2e-5 and rel(xg.grad, xr.grad) < 2e-5 and rel(wg.grad, wr.grad) < 3e-5
        outs[use_pk] = (yg.detach(), xg.grad.detach())
    assert rel(outs[True][0], outs[False][0].cpu()) < 1e-5 and rel(outs[True][1], outs[False][1].cpu()) < 1e-5
