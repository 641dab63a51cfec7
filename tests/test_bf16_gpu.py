"""GPU: the bf16-operand GEMM path (row g: BASELINE configs 3-5; north_star: "generated waveform RMS within 1e-3
bf16").  Two kinds of checks:
  * exactness of the kernels: a bf16-mode launch must equal the fp32 CPU convolution of the bf16-ROUNDED operands up
    to fp32 summation order (1e-5) -- rounding happens exactly once, on the way into the matrix cores;
  * the north_star tolerance: generator / inference waveforms in bf16 mode against the fp32 oracle, RMS <= 1e-3."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import fill_state_dict, keys_shapes_of

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16_mode():
    from vcvits_amd import ops
    ops.set_compute_dtype("bf16")
    yield ops
    ops.set_compute_dtype("f32")


def rb(t):
    """round to bf16 (nearest even) and back: what the kernels feed the matrix cores"""
    return t.to(torch.bfloat16).to(torch.float32)


def rel(a, b):
    return (a.detach().cpu().double() - b.double()).abs().max().item() / (b.abs().max().item() + 1e-30)


CASES = [
    # kind, B, C, M, T, K, s, pad, d, P, in_leaky
    ("conv", 2, 256, 512, 384, 5, 1, 2, 1, 1, False),     # WN in_layer
    ("conv", 2, 128, 128, 2048, 11, 1, 25, 5, 1, True),   # generator resblock, dilation 5
    ("conv", 2, 64, 64, 4096, 3, 1, 3, 3, 1, True),
    ("conv", 2, 32, 32, 8192, 7, 1, 3, 1, 1, True),
    ("conv", 3, 1025, 256, 384, 1, 1, 0, 1, 1, False),    # posterior `pre` (ragged channel tail)
    ("conv", 2, 256, 768, 204, 3, 1, 1, 1, 1, False),     # FFN
    ("conv", 2, 130, 100, 333, 5, 1, 2, 1, 1, False),     # ragged everything
    ("period", 2, 32, 128, 421, 5, 3, 2, 1, 13, False),
    ("period", 2, 128, 512, 141, 5, 3, 2, 1, 13, False),
    ("period", 2, 512, 1024, 47, 5, 3, 2, 1, 13, False),
    ("period", 2, 1024, 1024, 16, 5, 1, 2, 1, 13, False),  # U = 208: the 224-wide tile
    ("period", 2, 1024, 1024, 7, 5, 1, 2, 1, 37, False),   # U = 259: the 288-wide tile
    ("period", 2, 512, 1024, 304, 5, 3, 2, 1, 2, False),
    ("convT", 2, 256, 128, 256, 16, 8, 4, 1, 1, True),
    ("convT", 2, 128, 64, 2048, 4, 4, 0, 1, 1, True),
    ("convT", 2, 64, 32, 4096, 4, 2, 1, 1, 1, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-C%d-M%d-T%d-K%d-s%d-d%d-P%d" % (c[0], c[2], c[3], c[4], c[5], c[6], c[8], c[9]))
def test_bf16_conv_is_exact_on_rounded_operands(gpu, bf16_mode, case):
    ops = bf16_mode
    kind, B, C, M, T, K, s, pad, d, P, in_leaky = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    if kind == "convT":
        x, w, b = t(B, C, T), t(C, M, K) * (C * K / s) ** -0.5, t(M) * 0.1
    elif kind == "period":
        x, w, b = t(B, C, T, P), t(M, C, K, 1) * (C * K) ** -0.5, t(M) * 0.1
    else:
        x, w, b = t(B, C, T), t(M, C, K) * (C * K) ** -0.5, t(M) * 0.1
    # reference: fp32 CPU conv of the ROUNDED operands (leaky-ReLU of the input is applied before the rounding)
    xr = x.clone().requires_grad_(True)
    xin = rb(F.leaky_relu(x, 0.1)) if in_leaky else rb(x)
    xin.requires_grad_(True)
    wq = rb(w).requires_grad_(True)
    if kind == "convT":
        yr = F.conv_transpose1d(xin, wq, b, stride=s, padding=pad)
    elif kind == "period":
        yr = F.conv2d(xin, wq, b, stride=(s, 1), padding=(pad, 0))
    else:
        yr = F.conv1d(xin, wq, b, stride=s, padding=pad, dilation=d)
    before = dict(ops.LAUNCH_COUNTS)
    xg, wg, bg = (v.to(gpu).requires_grad_(True) for v in (x, w, b))
    if kind == "convT":
        yg = ops.conv_transpose1d(xg, wg, bg, stride=s, pad=pad, in_leaky=in_leaky, slope=0.1)
    else:
        yg = ops.conv1d(xg, wg, bg, stride=s, pad=pad, dil=d, in_leaky=in_leaky, slope=0.1)
    assert ops.LAUNCH_COUNTS["bf16"] == before["bf16"] + 1, "forward did not run on the bf16 kernel"
    assert rel(yg, yr.detach()) < 1e-5, ("y", rel(yg, yr.detach()))
    # within bf16 rounding of the UN-rounded fp32 result (sanity of the tolerance story: ~2^-9 per operand)
    with torch.no_grad():
        xf = F.leaky_relu(x, 0.1) if in_leaky else x
        yf = (F.conv_transpose1d(xf, w, b, stride=s, padding=pad) if kind == "convT" else
              F.conv2d(xf, w, b, stride=(s, 1), padding=(pad, 0)) if kind == "period" else
              F.conv1d(xf, w, b, stride=s, padding=pad, dilation=d))
    assert rel(yg, yf) < 2e-2
    # data gradient: dy and w rounded, fp32 sums; the input leaky-ReLU derivative is applied in fp32 afterwards
    gy = t(*yr.shape)
    gq = rb(gy)
    if kind == "convT":
        dxin = F.conv1d(gq, wq.detach(), stride=s, padding=pad)
    elif kind == "period":
        dxin = torch.autograd.grad(F.conv2d(xin, wq.detach(), None, stride=(s, 1), padding=(pad, 0)), xin, gq)[0]
    else:
        dxin = torch.autograd.grad(F.conv1d(xin, wq.detach(), None, stride=s, padding=pad, dilation=d), xin, gq)[0]
    if in_leaky:
        dxin = dxin * torch.where(x > 0, torch.ones_like(x), torch.full_like(x, 0.1))
    before = dict(ops.LAUNCH_COUNTS)
    yg.backward(gy.to(gpu))
    if kind == "convT" and s > 3:
        # the data gradient of a ConvTranspose with stride > 3 (the two first generator stages) stays on the fp32 kernel
        dxin = F.conv1d(gy, w, stride=s, padding=pad)
        if in_leaky:
            dxin = dxin * torch.where(x > 0, torch.ones_like(x), torch.full_like(x, 0.1))
    else:
        assert ops.LAUNCH_COUNTS["bf16"] > before["bf16"], "data gradient did not run on the bf16 kernel"
    assert rel(xg.grad, dxin) < 1e-5, ("dx", rel(xg.grad, dxin))
    # weight gradient: both activations rounded, fp32 sums (the leaky-ReLU of the input is applied before rounding)
    if kind == "convT":
        dwr = torch.autograd.grad(F.conv_transpose1d(xin, wq, None, stride=s, padding=pad), wq, gq)[0]
    elif kind == "period":
        dwr = torch.autograd.grad(F.conv2d(xin, wq, None, stride=(s, 1), padding=(pad, 0)), wq, gq)[0]
    else:
        dwr = torch.autograd.grad(F.conv1d(xin, wq, None, stride=s, padding=pad, dilation=d), wq, gq)[0]
    if not (kind == "convT" and s > 3):
        assert ops.LAUNCH_COUNTS["wgrad_bf16"] == before["wgrad_bf16"] + 1, "weight gradient did not run on the bf16 kernel"
        assert rel(wg.grad, dwr) < 2e-5, ("dw", rel(wg.grad, dwr))
    assert rel(bg.grad, gy.sum(dim=[i for i in range(gy.dim()) if i != 1])) < 1e-5
    # bit-reproducible: the split reduction is combined in a fixed order
    wg2 = w.to(gpu).requires_grad_(True)
    xg2 = x.to(gpu)
    y2 = (ops.conv_transpose1d(xg2, wg2, None, stride=s, pad=pad, in_leaky=in_leaky, slope=0.1) if kind == "convT" else
          ops.conv1d(xg2, wg2, None, stride=s, pad=pad, dil=d, in_leaky=in_leaky, slope=0.1))
    y2.backward(gy.to(gpu))
    if not (kind == "convT" and s > 3):
        assert torch.equal(wg2.grad, wg.grad), "weight gradient differs between two identical launches"


def _rms(a, b):
    return ((a.detach().cpu().double() - b.double()) ** 2).mean().sqrt().item()


@pytest.mark.parametrize("widths", ["reduced", "base", "48k"])
def test_generator_waveform_rms_bf16(gpu, bf16_mode, widths):
    """north_star: generated waveform RMS within 1e-3 (bf16) of the fp32 reference path.  HiFi-GAN Generator forward in
    bf16 mode against the fp32 CPU oracle, at reduced widths and at the two configs' real widths (32-frame segment),
    with fan-in-scaled weights (golden_util.fill_state_dict) so the waveform has content: signal RMS >= 0.1 is asserted
    (the reference's N(0, 0.01) initialisation gives a near-DC output of RMS 0.03, against which an absolute 1e-3 says
    little)."""
    from oracle import vits_oracle as O
    from golden_util import record_stats
    from vcvits_amd.model.generator import Generator
    ops = bf16_mode
    C, up = {"reduced": (16, 32), "base": (256, 512), "48k": (128, 512)}[widths]
    gen = Generator(C, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 4, 2], up, [16, 16, 4, 4])
    sd = fill_state_dict(keys_shapes_of(gen), seed=5)
    gen.load_state_dict(sd)
    rng = np.random.default_rng(7)
    z = torch.from_numpy(rng.standard_normal((2, C, 32)).astype(np.float32))
    with torch.no_grad():
        o_ref = O.generator_forward({"g." + k: v for k, v in sd.items()}, "g", z)
        before = ops.LAUNCH_COUNTS["bf16"]
        o = gen.to(gpu)(z.to(gpu))
        used = ops.LAUNCH_COUNTS["bf16"] - before
    assert o.shape == o_ref.shape == (2, 1, 16384)
    if widths != "reduced":
        assert used >= 60, "only %d launches of the generator ran on the bf16 kernel" % used
    sig = o_ref.pow(2).mean().sqrt().item()
    r = _rms(o, o_ref)
    record_stats("bf16wave", "generator/" + widths, rms_err=r, signal_rms=sig)
    assert sig >= 0.1, "test signal too weak to be meaningful: RMS %.3e" % sig
    assert r <= 1e-3, "waveform RMS error %.3e (signal RMS %.3e)" % (r, sig)
    # and it IS a different arithmetic: not bit-equal to the fp32 path at real widths
    if widths != "reduced":
        ops.set_compute_dtype("f32")
        with torch.no_grad():
            o32 = gen(z.to(gpu))
        ops.set_compute_dtype("bf16")
        assert _rms(o32, o_ref) < r


def test_infer_waveform_rms_bf16_48k(gpu, bf16_mode):
    """BASELINE configs[4] arithmetic: 48k widths, flow reverse + decoder in bf16 mode vs the fp32 oracle, 938 frames,
    B = 2.  Weights from golden_util.fill_state_dict (fan-in scaled, so the waveform has content: signal RMS >= 0.1 is
    asserted -- torch's default / N(0, 0.01) initialisation gives a near-DC output against which an absolute 1e-3 says
    little); north_star: "generated waveform RMS within 1e-3 bf16"."""
    from oracle import vits_oracle as O
    from golden_util import record_stats
    from vcvits_amd import configs
    from vcvits_amd.model.synthesizers.synthesizer_svc import SynthesizerSVC
    ops = bf16_mode
    cfg = configs.base_48k()
    d, m = cfg["data"], cfg["model"]
    net = SynthesizerSVC(d["filter_length"] // 2 + 1, 32, n_speakers=d["n_speakers"], **m).eval()
    sd0 = fill_state_dict(keys_shapes_of(net), seed=6)
    net.load_state_dict(sd0)
    sd = {"n." + k: v for k, v in sd0.items()}
    net = net.to(gpu)
    C, H, T, B = m["inter_channels"], m["hidden_channels"], 938, 2
    rng = np.random.default_rng(8)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    m_p, logs_p, noise = t(B, C, T), t(B, C, T) * 0.1 - 1.0, t(B, C, T)
    sid = torch.tensor([3, 5])
    mask = torch.ones(B, 1, T)
    with torch.no_grad():
        spk = net.emb_g(sid.to(gpu)).unsqueeze(-1)
        z_p = ops.prior_sample(m_p.to(gpu), logs_p.to(gpu), noise.to(gpu), 1.0)
        z = net.flow(z_p, mask.to(gpu), g=spk, reverse=True)
        count = lambda: ops.LAUNCH_COUNTS["bf16"] + ops.LAUNCH_COUNTS["bf16io"] + 2 * ops.LAUNCH_COUNTS.get("pair_fused", 0)
        before = count()
        o = net.dec(ops.mask_mul(z, mask.to(gpu).reshape(B, -1)))
        used = count() - before
        g = F.embedding(sid, sd["n.emb_g.weight"]).unsqueeze(-1)
        z_o = O.flow_forward(sd, "n.flow", m_p + noise * torch.exp(logs_p), mask, g, True, C, H, 5, 1, 4)
        o_o = O.generator_forward(sd, "n.dec", z_o * mask)
    assert used >= 60, "only %d launches of the decoder ran on the bf16 kernels" % used
    assert ops.LAUNCH_COUNTS["bf16io"] + 2 * ops.LAUNCH_COUNTS.get("pair_fused", 0) >= 76, "the decoder did not keep its activations in bf16"
    assert _rms(z, z_o) <= 2e-2 * z_o.pow(2).mean().sqrt().item()  # flow output within bf16 rounding of the fp32 path
    sig = o_o.pow(2).mean().sqrt().item()
    r = _rms(o, o_o)
    record_stats("bf16wave", "infer48k/B2", rms_err=r, signal_rms=sig)
    assert sig >= 0.1, "test signal too weak to be meaningful: RMS %.3e" % sig
    assert r <= 1e-3, "waveform RMS error %.3e (signal RMS %.3e)" % (r, sig)
    # batch rows are independent: row 1 of the B = 2 run equals the B = 1 run of that row (same kernels, other tiling)
    with torch.no_grad():
        z1 = net.flow(z_p[1:], mask[1:].to(gpu), g=spk[1:], reverse=True)
        o1 = net.dec(ops.mask_mul(z1, mask[1:].to(gpu).reshape(1, -1)))
    assert _rms(o1[0], o[1].cpu()) <= 2e-5


def test_grouped41_forward_bf16_operands(gpu):
    """DiscriminatorS's grouped k = 41 stride-4 convs in bf16 mode (vcv_grouped41_fwd_bf16: 4 taps x 4 channels per bf16 MFMA
    step): exact -- up to the accumulation order -- on operands that are bf16 numbers already; ragged time tiles."""
    from vcvits_amd import ops
    g0 = torch.Generator().manual_seed(41)
    for B, G, Tin in ((2, 4, 1000), (3, 16, 4096), (2, 64, 37 * 4 + 3)):
        x = torch.randn(B, G * 4, Tin, generator=g0).bfloat16().float().to(gpu)
        w = (torch.randn(G * 16, 4, 41, generator=g0) * 0.1).bfloat16().float().to(gpu)
        b = torch.randn(G * 16, generator=g0).to(gpu)
        ops.set_compute_dtype("f32")
        ref = ops.conv_forward(x, w, b, stride=4, pad=20, groups=G, out_act=ops.ACT_LEAKY)
        try:
            ops.set_compute_dtype("bf16")
            got = ops.conv_forward(x, w, b, stride=4, pad=20, groups=G, out_act=ops.ACT_LEAKY)
        finally:
            ops.set_compute_dtype("f32")
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (B, G, Tin, err)
        # data gradient (vcv_grouped41_dgrad_bf16: K of the MFMA = the group's 16 output channels): exact without the fused
        # activation derivative; with it the masked gradient 0.1 * dy is rounded to bf16 (2^-9)
        dy = torch.randn(ref.shape, generator=g0).bfloat16().float().to(gpu)
        for tf, aux, tol in ((ops.TF_NONE, None, 1e-5), (ops.TF_DLEAKY, ref, 6e-3)):
            ops.set_compute_dtype("f32")
            dref = ops.conv_dgrad(dy, w, x.shape, stride=4, pad=20, groups=G, in_tf=tf, xaux=aux, slope=0.1)
            try:
                ops.set_compute_dtype("bf16")
                dgot = ops.conv_dgrad(dy, w, x.shape, stride=4, pad=20, groups=G, in_tf=tf, xaux=aux, slope=0.1)
            finally:
                ops.set_compute_dtype("f32")
            derr = float((dgot - dref).abs().max() / dref.abs().max())
            assert derr < tol, ("dgrad", B, G, Tin, tf, derr)
            ops.set_compute_dtype("f32")
            wref = ops.conv_wgrad(dy, x, w.shape, stride=4, pad=20, groups=G, a_tf=tf, aaux=aux, slope=0.1)
            try:
                ops.set_compute_dtype("bf16")
                wgot = ops.conv_wgrad(dy, x, w.shape, stride=4, pad=20, groups=G, a_tf=tf, aaux=aux, slope=0.1)
            finally:
                ops.set_compute_dtype("f32")
            werr = float((wgot - wref).abs().max() / wref.abs().max())
            assert werr < (2e-5 if tf == ops.TF_NONE else tol), ("wgrad", B, G, Tin, tf, werr)


@pytest.mark.parametrize("shape", [(2, 64, 96, 3000, 1), (2, 48, 40, 3000, 3), (3, 32, 64, 2500, 5), (2, 36, 33, 2000, 7),
                                   (2, 64, 32, 2200, 11), (2, 32, 32, 1800, 16), (2, 50, 64, 2000, 5)],
                         ids=lambda s: "B%d-C%d-M%d-T%d-K%d" % s)
def test_wgrad_bf16_split_reduction_every_finish_variant(gpu, bf16_mode, shape):
    """The split-reduction finishing pass (wgrad_bf16_finish4_kernel: z-lane groups 4 / 8 / 16 / 32 x tap buckets
    1 / 3 / 5 / 8 / 11 / 16, and the 4-byte kernel for channel counts that are not a multiple of 4) at forced split counts:
    every count gives the fp32 CPU gradient of the rounded operands, and a count is bit-reproducible."""
    from vcvits_amd import _lib
    ops = bf16_mode
    B, C, M, T, K = shape
    rng = np.random.default_rng(B * 1000 + C * 10 + K)
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    x, dy = t(B, C, T), t(B, M, T)
    pad = K // 2 if K % 2 else 0
    if not K % 2:
        dy = dy[:, :, :T - K + 1].contiguous()
    wq = torch.zeros(M, C, K, requires_grad=True)
    ref = torch.autograd.grad(F.conv1d(rb(x), wq, None, padding=pad), wq, rb(dy))[0]
    xg, dyg = x.to(gpu), dy.to(gpu)
    L = _lib.lib()
    try:
        for z in (1, 3, 4, 7, 8, 13, 16, 29, 64):
            L.vcv_wgrad_bf16_set_force(-1, z)
            before = ops.LAUNCH_COUNTS["wgrad_bf16"]
            got = ops.conv_wgrad(dyg, xg, (M, C, K), pad=pad)
            assert ops.LAUNCH_COUNTS["wgrad_bf16"] == before + 1
            assert rel(got, ref) < 2e-5, (z, rel(got, ref))
            again = ops.conv_wgrad(dyg, xg, (M, C, K), pad=pad)
            assert torch.equal(got, again), "split count %d is not bit-reproducible" % z
            # accumulate onto an existing gradient with a scale
            acc = got.clone()
            ops.conv_wgrad(dyg, xg, (M, C, K), pad=pad, out=acc, alpha=0.5)
            assert rel(acc, 1.5 * ref) < 2e-5, (z, "accumulate")
    finally:
        L.vcv_wgrad_bf16_set_force(-1, -1)
