"""GPU: one conv pair of a ResBlock1 as ONE launch over fp16 activations (csrc/resblock_pair.hip, ops.resblock_pair_x16)
against (i) the two-launch path it replaces (vcv_conv_bf16io_*: same operands, same roundings -- only the fp32 summation
order inside a conv differs) and (ii) a CPU restatement of the arithmetic in fp32 on the same rounded operands.
Reference: vits/model/modules.py:186-222 (ResBlock1.forward) under fp16 autocast (train.py:104-106)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SLOPE = 0.1


def _inputs(B, C, K, T, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, C, T, generator=g) * 0.7).half()
    w1 = torch.randn(C, C, K, generator=g) * (C * K) ** -0.5
    w2 = torch.randn(C, C, K, generator=g) * (C * K) ** -0.5
    b1 = torch.randn(C, generator=g) * 0.1
    b2 = torch.randn(C, generator=g) * 0.1
    return x, w1, b1, w2, b2


def _cpu_reference(x, w1, b1, w2, b2, dil):
    """The pair in fp32 on the operands the kernels see: bf16(leaky(x)), bf16 weights, xt rounded to bf16, fp32 sums."""
    K = w1.shape[2]
    bf = lambda t: t.bfloat16().float()
    xl = bf(F.leaky_relu(x.float(), SLOPE))
    xt = F.conv1d(xl, bf(w1), b1, dilation=dil, padding=(K - 1) * dil // 2)
    xt = bf(F.leaky_relu(xt, SLOPE))
    return F.conv1d(xt, bf(w2), b2, padding=(K - 1) // 2) + x.float()


@pytest.mark.parametrize("C,K,dil,T,B", [(32, 3, 1, 1000, 3), (32, 7, 3, 2056, 2), (32, 11, 5, 1504, 3), (32, 11, 1, 488, 2),
                                         (64, 3, 5, 1000, 2), (64, 3, 1, 760, 3), (64, 3, 3, 256, 2),
                                         (64, 7, 3, 760, 3), (64, 7, 5, 1248, 2), (64, 11, 5, 1000, 2), (64, 11, 1, 504, 3)])
def test_fused_pair_equals_two_launches_and_the_cpu_arithmetic(gpu, C, K, dil, T, B):
    from vcvits_amd import ops
    from vcvits_amd._lib import ACT_LEAKY
    x, w1, b1, w2, b2 = _inputs(B, C, K, T, seed=C * 100 + K * 10 + dil)
    xg, w1g, b1g, w2g, b2g = (t.to(gpu) for t in (x, w1, b1, w2, b2))
    assert ops.resblock_pair_supported(xg, w1g, w2g, dil)
    ops.set_compute_dtype("bf16")
    try:
        before = ops.LAUNCH_COUNTS.get("pair_fused", 0)
        y = ops.resblock_pair_x16(xg, w1g, b1g, w2g, b2g, dil, slope=SLOPE)
        assert ops.LAUNCH_COUNTS["pair_fused"] == before + 1 and y.dtype == torch.float16 and y.shape == x.shape
        xt = ops.conv_forward_x16(xg, w1g, b1g, pad=(K - 1) * dil // 2, dil=dil, in_leaky=True, out_act=ACT_LEAKY, slope=SLOPE,
                                  out_dtype=torch.bfloat16)
        y2 = ops.conv_forward_x16(xt, w2g, b2g, pad=(K - 1) // 2, dil=1, res=xg, out_dtype=torch.float16)
        # the block's last pair: y_acc = acc + scale * pair(x)
        acc0 = (torch.randn(B, C, T, generator=torch.Generator().manual_seed(5)) * 0.3).half().to(gpu)
        ya = ops.resblock_pair_x16(xg, w1g, b1g, w2g, b2g, dil, slope=SLOPE, out=acc0.clone(), accumulate=True, post_scale=1.0 / 3)
        ys = ops.resblock_pair_x16(xg, w1g, b1g, w2g, b2g, dil, slope=SLOPE, post_scale=1.0 / 3)
    finally:
        ops.set_compute_dtype("f32")
    ref = _cpu_reference(x, w1, b1, w2, b2, dil)
    scale = ref.abs().max().item()
    rms = ref.pow(2).mean().sqrt().item()
    yf, y2f = y.float().cpu(), y2.float().cpu()
    # (ii) against the fp32 CPU arithmetic: one fp16 rounding of the result (2^-11 relative) + the occasional xt element whose
    # fp32 sum lands on the other side of a bf16 rounding boundary
    err = (yf - ref).abs()
    assert err.max().item() <= 6e-3 * scale, (err.max().item(), scale)
    assert err.pow(2).mean().sqrt().item() <= 6e-4 * rms
    # (i) against the two launches: the same statistics, and identical on all but a small share of the elements
    d = (yf - y2f).abs()
    assert d.max().item() <= 6e-3 * scale
    assert (d > 0).float().mean().item() <= 0.12, (d > 0).float().mean().item()
    # accumulate / post-scale epilogues
    want_a = acc0.float().cpu() + ref / 3
    assert (ya.float().cpu() - want_a).abs().max().item() <= 6e-3 * max(scale, 1.0)
    assert (ys.float().cpu() - ref / 3).abs().max().item() <= 6e-3 * scale


@pytest.mark.parametrize("C,K,dil,T,B,grid", [(64, 11, 3, 2504, 2, 3), (64, 7, 5, 2000, 3, 2), (64, 11, 1, 1520, 2, 1),
                                              (32, 11, 5, 2504, 2, 3), (64, 3, 3, 2000, 2, 2)])
def test_persistent_workgroups_walk_many_tiles(gpu, C, K, dil, T, B, grid):
    """A grid of 1 - 3 workgroups walks 12 - 30 tiles each: the weight ring of the streamed variants (64 channels, K >= 7)
    runs on across tiles -- its slab positions shift by 2 K mod 4 per tile and the first slabs of the NEXT tile are copied
    under the epilogue of the current one -- and the next tile's input prefetch rides over both convs.  Bit-identical to the
    same launch on the default grid (one tile per workgroup at these sizes)."""
    from vcvits_amd import ops
    from vcvits_amd._lib import lib
    x, w1, b1, w2, b2 = _inputs(B, C, K, T, seed=C + K + dil)
    xg, w1g, b1g, w2g, b2g = (t.to(gpu) for t in (x, w1, b1, w2, b2))
    ops.set_compute_dtype("bf16")
    try:
        y0 = ops.resblock_pair_x16(xg, w1g, b1g, w2g, b2g, dil, slope=SLOPE)
        assert lib().vcv_tuning_set(b"pair_grid", grid) == 0
        try:
            y1 = ops.resblock_pair_x16(xg, w1g, b1g, w2g, b2g, dil, slope=SLOPE)
            acc0 = (torch.randn(B, C, T, generator=torch.Generator().manual_seed(7)) * 0.3).half().to(gpu)
            ya = ops.resblock_pair_x16(xg, w1g, b1g, w2g, b2g, dil, slope=SLOPE, out=acc0.clone(), accumulate=True, post_scale=0.5)
        finally:
            lib().vcv_tuning_set(b"pair_grid", 0)
        yb = ops.resblock_pair_x16(xg, w1g, b1g, w2g, b2g, dil, slope=SLOPE, out=acc0.clone(), accumulate=True, post_scale=0.5)
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype("f32")
    assert torch.equal(y0, y1), float((y0.float() - y1.float()).abs().max())
    assert torch.equal(ya, yb)
    ref = _cpu_reference(x, w1, b1, w2, b2, dil)
    assert (y1.float().cpu() - ref).abs().max().item() <= 6e-3 * ref.abs().max().item()


def test_fused_pair_declines_what_does_not_fit(gpu):
    from vcvits_amd import ops
    x = torch.zeros(1, 64, 512, dtype=torch.float16, device=gpu)
    w = torch.zeros(64, 64, 5, device=gpu)
    assert not ops.resblock_pair_supported(x, w, w, 1)          # kernel sizes 3 / 7 / 11 only
    assert not ops.resblock_pair_supported(x, torch.zeros(64, 64, 11, device=gpu), torch.zeros(64, 64, 11, device=gpu), 7)  # dilation <= 5
    x2 = torch.zeros(1, 32, 500, dtype=torch.float16, device=gpu)  # rows of a multiple of eight elements only
    assert not ops.resblock_pair_supported(x2, torch.zeros(32, 32, 3, device=gpu), torch.zeros(32, 32, 3, device=gpu), 1)
    x3 = torch.zeros(1, 128, 512, dtype=torch.float16, device=gpu)
    assert not ops.resblock_pair_supported(x3, torch.zeros(128, 128, 3, device=gpu), torch.zeros(128, 128, 3, device=gpu), 1)
