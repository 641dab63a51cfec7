"""GPU parity of the conv family (vcv_conv_gemm / vcv_conv_wgrad) against torch CPU convs.

torch CPU F.conv1d / F.conv2d / F.conv_transpose1d + CPU autograd are the fp32 reference of the
same op (they are what the reference's modules call: modules.py:126-143,190-201,
discriminator.py:18-25,53-61).  Tolerance: max|diff| <= 2e-5 * max|ref| (fp32, different
summation order only).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _cmp(name, got, ref, tol=TOL):
    got = got.detach().cpu()
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item() / scale
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert err <= tol, "%s: rel err %.3e" % (name, err)
    return err


CONV_CASES = [
    # B, C, M, T, K, stride, pad, dil, groups
    (2, 16, 32, 50, 5, 1, 2, 1, 1),
    (2, 256, 512, 32, 7, 1, 3, 1, 1),
    (3, 64, 64, 200, 3, 1, 1, 1, 1),
    (2, 64, 64, 300, 11, 1, 25, 5, 1),
    (2, 32, 32, 1000, 7, 1, 9, 3, 1),
    (2, 1025, 96, 70, 1, 1, 0, 1, 1),
    (2, 96, 192, 77, 1, 1, 0, 1, 1),
    (2, 1, 16, 500, 15, 1, 7, 1, 1),
    (2, 16, 64, 512, 41, 4, 20, 1, 4),
    (2, 64, 256, 260, 41, 4, 20, 1, 16),
    (2, 1024, 1024, 40, 41, 4, 20, 1, 256),
    (2, 1024, 1024, 24, 5, 1, 2, 1, 1),
    (2, 1024, 1, 24, 3, 1, 1, 1, 1),
    (2, 32, 1, 700, 7, 1, 3, 1, 1),
    (1, 130, 70, 33, 3, 1, 1, 1, 1),
    # short sequences: the batch is folded into the column dimension (pooled DiscriminatorS scales)
    (4, 64, 96, 5, 5, 1, 2, 1, 1),
    (3, 128, 64, 33, 5, 1, 2, 1, 1),
    (2, 1024, 1024, 9, 5, 1, 2, 1, 1),
    (3, 256, 512, 384, 1, 1, 0, 1, 1),  # pointwise conv on the DMA kernel
    (2, 1025, 256, 204, 1, 1, 0, 1, 1),
    (4, 64, 96, 64, 5, 1, 2, 1, 1),   # folded forward / data gradient, unfolded weight gradient
    # few output tiles: reduction split over several blocks per tile + finishing pass
    (16, 512, 512, 20, 5, 1, 2, 1, 1),
    (2, 256, 256, 256, 11, 1, 5, 1, 1),
    (2, 256, 192, 200, 3, 1, 3, 3, 1),
    (1, 300, 130, 150, 7, 1, 3, 1, 1),
    # one output channel, LDS-staged: rows of 2..5 positions (pooled DiscriminatorS heads), several position
    # tiles with a dilated halo, a row that is not a multiple of the tile
    (64, 1024, 1, 2, 3, 1, 1, 1, 1),
    (3, 1024, 1, 5, 3, 1, 1, 1, 1),
    (3, 40, 1, 1000, 7, 1, 9, 3, 1),
    (2, 32, 1, 8192, 7, 1, 3, 1, 1),
    (5, 256, 1, 111, 3, 1, 1, 1, 1),
    (2, 24, 1, 300, 5, 1, 0, 1, 1),
    (3, 1, 16, 3000, 15, 1, 7, 1, 1),  # one input channel: several row chunks per batch element
    (3, 48, 1, 70, 5, 1, 4, 2, 1),     # one output channel: three channel sub-rows per workgroup, dilated halo
    (2, 1, 17, 300, 15, 1, 7, 1, 1),   # one input channel: 255 of the 256 weight lanes
    (2, 1, 18, 300, 15, 1, 7, 1, 1),   # ... 270 weights: the row-per-workgroup kernel
    (2, 1, 40, 257, 5, 1, 2, 1, 1),
    (5, 20, 1, 1, 3, 1, 1, 1, 1),      # a single position per row
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d_fwd_bwd(gpu, case):
    from vcvits_amd import ops
    B, C, M, T, K, s, p, d, g = case
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, C, T, generator=gen)
    w = torch.randn(M, C // g, K, generator=gen) / (C // g * K) ** 0.5
    b = torch.randn(M, generator=gen)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv1d(xr, wr, br, stride=s, padding=p, dilation=d, groups=g)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xg, wg, bg = (t.clone().to(gpu).requires_grad_(True) for t in (x, w, b))
    yg = ops.conv1d(xg, wg, bg, stride=s, pad=p, dil=d, groups=g)
    yg.backward(gy.to(gpu))
    _cmp("y", yg, yr.detach())
    _cmp("dx", xg.grad, xr.grad)
    _cmp("dw", wg.grad, wr.grad)
    _cmp("db", bg.grad, br.grad)


PERIOD_CASES = [
    # B, C, M, H, P, K, stride, pad
    (2, 1, 32, 100, 2, 5, 3, 2),
    (2, 32, 128, 67, 3, 5, 3, 2),
    (2, 128, 512, 23, 5, 5, 3, 2),
    (2, 512, 1024, 9, 37, 5, 3, 2),
    (2, 1024, 1024, 6, 37, 5, 1, 2),
    (2, 1024, 1, 6, 11, 3, 1, 1),
    (3, 1024, 1, 51, 2, 3, 1, 1),
    (3, 1024, 1, 3, 37, 3, 1, 1),
    (2, 1, 32, 700, 37, 5, 3, 2),
    (2, 1, 32, 1400, 3, 5, 3, 2),
    (2, 64, 1, 9, 5, 3, 1, 1),
    (2, 1, 16, 40, 7, 15, 1, 7),
    (2, 32, 128, 911, 2, 5, 3, 2),
]


@pytest.mark.parametrize("case", PERIOD_CASES)
def test_period_conv2d_fwd_bwd(gpu, case):
    from vcvits_amd import ops
    from vcvits_amd._lib import ACT_LEAKY
    B, C, M, H, P, K, s, p = case
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, C, H, P, generator=gen)
    w = torch.randn(M, C, K, 1, generator=gen) / (C * K) ** 0.5
    b = torch.randn(M, generator=gen)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.leaky_relu(F.conv2d(xr, wr, br, stride=(s, 1), padding=(p, 0)), 0.1)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xg, wg, bg = (t.clone().to(gpu).requires_grad_(True) for t in (x, w, b))
    yg = ops.conv1d(xg, wg, bg, stride=s, pad=p, out_act=ACT_LEAKY, slope=0.1)
    yg.backward(gy.to(gpu))
    _cmp("y", yg, yr.detach())
    _cmp("dx", xg.grad, xr.grad)
    _cmp("dw", wg.grad, wr.grad)
    _cmp("db", bg.grad, br.grad)


CONVT_CASES = [
    # B, Cin, Cout, T, K, stride, pad
    (2, 512, 256, 32, 16, 8, 4),
    (2, 256, 128, 64, 16, 8, 4),
    (2, 128, 64, 100, 4, 4, 0),
    (2, 64, 32, 300, 4, 2, 1),
    (2, 20, 12, 17, 3, 1, 1),
    (16, 512, 256, 32, 16, 8, 4),  # the first generator stage at the bench's batch: folded forward and data gradient
    (3, 40, 24, 17, 4, 4, 0),
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv_transpose1d_fwd_bwd(gpu, case):
    from vcvits_amd import ops
    B, C, M, T, K, s, p = case
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, C, T, generator=gen)
    w = torch.randn(C, M, K, generator=gen) / (C * K / s) ** 0.5
    b = torch.randn(M, generator=gen)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv_transpose1d(F.leaky_relu(xr, 0.1), wr, br, stride=s, padding=p)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xg, wg, bg = (t.clone().to(gpu).requires_grad_(True) for t in (x, w, b))
    yg = ops.conv_transpose1d(xg, wg, bg, stride=s, pad=p, in_leaky=True, slope=0.1)
    yg.backward(gy.to(gpu))
    _cmp("y", yg, yr.detach())
    _cmp("dx", xg.grad, xr.grad)
    _cmp("dw", wg.grad, wr.grad)
    _cmp("db", bg.grad, br.grad)


def test_resblock_style_fusions(gpu):
    """in_leaky + residual epilogue and tanh output, as the HiFi-GAN blocks use them."""
    from vcvits_amd import ops
    from vcvits_amd._lib import ACT_TANH
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(2, 48, 120, generator=gen)
    w1 = torch.randn(48, 48, 7, generator=gen) * 0.05
    w2 = torch.randn(1, 48, 7, generator=gen) * 0.05
    b1 = torch.randn(48, generator=gen)
    xr, w1r, w2r, b1r = (t.clone().requires_grad_(True) for t in (x, w1, w2, b1))
    hr = F.conv1d(F.leaky_relu(xr, 0.1), w1r, b1r, padding=9, dilation=3) + xr
    yr = torch.tanh(F.conv1d(F.leaky_relu(hr, 0.01), w2r, None, padding=3))
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xg, w1g, w2g, b1g = (t.clone().to(gpu).requires_grad_(True) for t in (x, w1, w2, b1))
    hg = ops.conv1d(xg, w1g, b1g, pad=9, dil=3, in_leaky=True, slope=0.1, res=xg)
    yg = ops.conv1d(hg, w2g, None, pad=3, in_leaky=True, slope=0.01, out_act=ACT_TANH)
    yg.backward(gy.to(gpu))
    _cmp("y", yg, yr.detach())
    _cmp("dx", xg.grad, xr.grad)
    _cmp("dw1", w1g.grad, w1r.grad)
    _cmp("dw2", w2g.grad, w2r.grad)
    _cmp("db1", b1g.grad, b1r.grad)


@pytest.mark.parametrize("case", [(16, 256, 6144), (2, 8, 96), (32, 256, 1024), (1, 300, 33)])
def test_one_frame_pointwise_layer(gpu, case):
    """Speaker-conditioning layers: Conv1d(C, M, 1) on [B, C, 1] (vcv_linear_t1_*), forward and all gradients."""
    from vcvits_amd import ops
    B, C, M = case
    gen = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, C, 1, generator=gen)
    w = torch.randn(M, C, 1, generator=gen) / C ** 0.5
    b = torch.randn(M, generator=gen)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv1d(xr, wr, br)
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xg, wg, bg = (t.clone().to(gpu).requires_grad_(True) for t in (x, w, b))
    yg = ops.conv1d(xg, wg, bg)
    yg.backward(gy.to(gpu))
    _cmp("y", yg, yr.detach())
    _cmp("dx", xg.grad, xr.grad)
    _cmp("dw", wg.grad, wr.grad)
    _cmp("db", bg.grad, br.grad)
