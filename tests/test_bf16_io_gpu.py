"""GPU: 16-bit ACTIVATIONS in HBM (vcv_conv_bf16io_*, conv_pk_io*.hip) -- the decoder's inference pass in bf16 mode keeps
every conv <-> conv tensor in 16 bits, as the reference's fp16 autocast does (train.py:104-106, synthesizer_svc.py:108): the
residual stream as fp16, the tensor between the two convs of a ResBlock pair as the bf16 matrix-core operand itself.
  * kernel exactness: a launch equals the fp32 CPU convolution of the SAME bf16 inputs (fp32 accumulate, fp32 epilogue)
    rounded once to bf16 -- compared before the rounding to 1e-5 would need the fp32 value, so the check is: within half a
    bf16 ulp (+ fp32 summation-order slack) of the fp32 reference, and bit-equal to its rounding on all but a few ties;
  * the north_star tolerance: decoder waveform RMS <= 1e-3 against the fp32 oracle with every activation stored in bf16."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import fill_state_dict, keys_shapes_of, record_stats

pytestmark = pytest.mark.gpu


@pytest.fixture
def bf16_mode():
    from vcvits_amd import ops
    ops.set_compute_dtype("bf16")
    yield ops
    ops.set_compute_dtype("f32")


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rs(t, dtype):
    """round to a storage dtype and back"""
    return t.to(dtype).to(torch.float32)


def check_rounded(name, got_bf16, ref_f32):
    """got (bf16 tensor from the GPU) vs the fp32 reference before its rounding: every element within half a bf16 ulp of
    the reference (2^-9 relative) plus fp32 summation-order slack, and equal to the reference's own rounding except where
    the reference sits within that slack of a rounding boundary."""
    got = got_bf16.detach().float().cpu().double()
    ref = ref_f32.double()
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    ulp = 2.0 ** -8 if got_bf16.dtype == torch.bfloat16 else 2.0 ** -11
    bound = ref.abs() * ulp + 2e-5 * scale   # one ulp of the storage type at the element's size, generous at tiny values
    assert bool((err <= bound).all()), (name, (err / (ref.abs() + 1e-3 * scale)).max().item())
    same = (got == rs(ref_f32, got_bf16.dtype).double()).double().mean().item()
    # (fp32 summation order moves a result across a rounding boundary of the 8x finer fp16 grid 8x as often)
    assert same > (0.99 if got_bf16.dtype == torch.bfloat16 else 0.93), (name, "only %.4f of the elements equal the rounded reference" % same)


CASES = [
    # C, M, T, K, dil, in_leaky, out_act_leaky, res, accumulate
    (32, 32, 8192, 3, 1, True, True, False, False),
    (32, 32, 8192, 11, 5, True, True, False, False),
    (32, 32, 8192, 7, 1, False, False, True, False),
    (32, 32, 4104, 7, 1, False, False, True, True),      # ragged position tile
    (64, 64, 4096, 11, 3, True, True, False, False),
    (64, 64, 4096, 3, 1, False, False, True, True),
    (128, 128, 2048, 7, 5, True, True, False, False),
    (128, 128, 2048, 11, 1, False, False, True, False),
    (256, 256, 944, 3, 3, True, True, False, False),
    (256, 256, 938, 7, 1, False, False, True, True),      # rows of 938 = 2 * 469 elements: the scalar-store epilogue
    (130, 100, 1000, 5, 1, True, False, False, False),    # ragged channels both ways
]


STORAGE = {"bb": (torch.bfloat16, torch.bfloat16), "hb": (torch.float16, torch.bfloat16), "bh": (torch.bfloat16, torch.float16),
           "hh": (torch.float16, torch.float16)}


@pytest.mark.parametrize("storage", ["bb", "hb", "bh", "hh"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "C%d-M%d-T%d-K%d-d%d-l%d-a%d-r%d-acc%d" % tuple(int(v) for v in c))
def test_conv_x16_matches_rounded_reference(gpu, bf16_mode, case, storage):
    ops = bf16_mode
    xdt, ydt = STORAGE[storage]
    from vcvits_amd._lib import ACT_LEAKY, ACT_NONE
    C, M, T, K, dil, in_leaky, act, use_res, acc = case
    B = 2
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    x, w, b = rs(t(B, C, T), xdt), t(M, C, K) * (C * K) ** -0.5, t(M) * 0.1
    pad = dil * (K - 1) // 2
    res = rs(t(B, M, T), ydt) if use_res else None
    y0 = rs(t(B, M, T), ydt) if acc else None
    ps = 1.0 / 3.0 if acc else 0.0
    xin = rb(F.leaky_relu(x, 0.1)) if in_leaky else rb(x)  # the matrix-core operand is bf16 whatever the storage
    ref = F.conv1d(xin, rb(w), b, padding=pad, dilation=dil)
    if act:
        ref = F.leaky_relu(ref, 0.1)
    if use_res:
        ref = ref + res
    if acc:
        ref = ref * np.float32(ps) + y0
    before = ops.LAUNCH_COUNTS["bf16io"]
    out = y0.to(ydt).to(gpu) if acc else None
    y = ops.conv_forward_x16(x.to(xdt).to(gpu), w.to(gpu), b.to(gpu), pad=pad, dil=dil, in_leaky=in_leaky,
                             out_act=ACT_LEAKY if act else ACT_NONE, slope=0.1,
                             res=res.to(ydt).to(gpu) if use_res else None, out=out, accumulate=acc, post_scale=ps, out_dtype=ydt)
    assert ops.LAUNCH_COUNTS["bf16io"] == before + 1
    assert y.dtype == ydt and tuple(y.shape) == (B, M, T)
    check_rounded("y", y, ref)


@pytest.mark.parametrize("case", [(256, 128, 944, 16, 8, 4), (128, 64, 2048, 16, 8, 4), (64, 32, 4096, 4, 4, 0), (32, 16, 4096, 4, 2, 1), (64, 32, 4096, 4, 2, 1), (128, 64, 1000, 4, 4, 0),
                                  (512, 256, 938, 16, 8, 4)],
                         ids=lambda c: "C%d-M%d-T%d-K%d-s%d" % c[:5])
@pytest.mark.parametrize("merged", [True, False], ids=["merged", "phased"])
@pytest.mark.parametrize("storage", ["bb", "hh"])
def test_convT_x16_matches_rounded_reference(gpu, bf16_mode, case, storage, merged):
    """merged: all output phases as rows of one launch (VcvConvArgs.ms, interleaving epilogue); phased: one launch phase per
    output residue."""
    ops = bf16_mode
    ops._CONVT_MERGED[0] = merged
    xdt, ydt = STORAGE[storage]
    C, M, T, K, s, pad = case
    if M < 32:
        pytest.skip("fewer than 32 output channels: not a tile of the packed kernels")
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    x, w, b = rs(t(2, C, T), xdt), t(C, M, K) * (C * K / s) ** -0.5, t(M) * 0.1
    ref = F.conv_transpose1d(rb(F.leaky_relu(x, 0.1)), rb(w), b, stride=s, padding=pad)
    before = ops.LAUNCH_COUNTS["bf16io"]
    y = ops.convT_forward_x16(x.to(xdt).to(gpu), w.to(gpu), b.to(gpu), stride=s, pad=pad, in_leaky=True, slope=0.1, out_dtype=ydt)
    ops._CONVT_MERGED[0] = True
    assert y.dtype == ydt
    assert ops.LAUNCH_COUNTS["bf16io"] == before + 1
    check_rounded("y", y, ref)


def test_conv_m1_x16_and_casts(gpu, bf16_mode):
    ops = bf16_mode
    from vcvits_amd._lib import ACT_TANH
    rng = np.random.default_rng(3)
    t = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    for dt in (torch.bfloat16, torch.float16):
        for (C, T, K) in [(32, 16384, 7), (32, 4102, 7), (64, 2048, 3), (16, 1000, 5)]:
            x, w = rs(t(2, C, T), dt), t(1, C, K) * (C * K) ** -0.5
            ref = torch.tanh(F.conv1d(F.leaky_relu(x, 0.01), w, None, padding=(K - 1) // 2))
            y = ops.conv_m1_x16(x.to(dt).to(gpu), w.to(gpu), None, pad=(K - 1) // 2, in_leaky=True, slope=0.01, out_act=ACT_TANH)
            assert y.dtype == torch.float32
            assert (y.cpu() - ref).abs().max().item() < 2e-6, (dt, C, T, K)
        v = t(3, 5, 1001) * 10
        v[0, 0, :4] = torch.tensor([1e6, -1e6, 70000.0, 65504.0])  # fp16: clamped to the finite range, not inf
        vb = ops.cast_x16(v.to(gpu), dt)
        want = v.clamp(-65504.0, 65504.0).to(dt) if dt == torch.float16 else v.to(dt)
        assert torch.equal(vb.cpu(), want)
        assert torch.equal(ops.cast_f32(vb).cpu(), want.float())


def test_other_families_refuse_bf16_activations(gpu):
    """io / post_scale are honoured by vcv_conv_bf16io_* only: every other entry point returns VCV_EINVAL."""
    import ctypes
    from vcvits_amd import _lib
    L = _lib.lib()
    x = torch.zeros(2, 64, 512, device=gpu)
    w = torch.zeros(64, 64, 3, device=gpu)
    y = torch.zeros(2, 64, 512, device=gpu)
    for io, ps in ((3, 0.0), (15, 0.0), (0, 0.5)):
        a = _lib.VcvConvArgs()
        a.x, a.w, a.y = _lib.ptr(x), _lib.ptr(w), _lib.ptr(y)
        a.B, a.G, a.Cg, a.Mg, a.Tin, a.Tout, a.P, a.K = 2, 1, 64, 64, 512, 512, 1, 3
        a.s, a.dj, a.off, a.os, a.oo, a.phases, a.Q, a.a_mode = 1, 1, -1, 1, 0, 1, 512, 0
        a.alpha, a.slope, a.io, a.post_scale = 1.0, 0.1, io, ps
        plan = (ctypes.c_int64 * 3)()
        for name in ("vcv_conv_bf16_plan", "vcv_conv_pk_plan", "vcv_conv_x3_plan", "vcv_conv_dma_plan"):
            assert getattr(L, name)(ctypes.byref(a), 0, plan) != 0, name
        assert L.vcv_conv_gemm(ctypes.byref(a), _lib.stream()) != 0


@pytest.mark.parametrize("widths", ["base", "48k"])
def test_generator_waveform_rms_bf16_activations(gpu, bf16_mode, widths):
    """north_star: generated waveform RMS within 1e-3 (bf16).  128 frames (the bf16-activation path needs >= 96 positions per
    launch; the 32-frame training segment keeps fp32 activations), fan-in-scaled weights, signal RMS >= 0.1 asserted."""
    from oracle import vits_oracle as O
    from vcvits_amd.model.generator import Generator
    ops = bf16_mode
    C, up = {"base": (256, 512), "48k": (128, 512)}[widths]
    gen = Generator(C, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 4, 2], up, [16, 16, 4, 4])
    sd = fill_state_dict(keys_shapes_of(gen), seed=5)
    gen.load_state_dict(sd)
    rng = np.random.default_rng(7)
    z = torch.from_numpy(rng.standard_normal((2, C, 128)).astype(np.float32))
    with torch.no_grad():
        o_ref = O.generator_forward({"g." + k: v for k, v in sd.items()}, "g", z)
        before = dict(ops.LAUNCH_COUNTS)
        o = gen.to(gpu)(z.to(gpu))
        # (a fused ResBlock pair launch -- the 32- / 64-channel stages, resblock_pair.hip -- stands for two of the 72 convs)
        used = ops.LAUNCH_COUNTS["bf16io"] - before["bf16io"] + 2 * (ops.LAUNCH_COUNTS.get("pair_fused", 0) - before.get("pair_fused", 0))
        assert used == 4 + 72, "%d launches on the bf16-activation kernel (4 transposed convs + 72 ResBlock convs expected)" % used
        ops.set_bf16_activations(False)
        try:
            o_f32act = gen(z.to(gpu))
        finally:
            ops.set_bf16_activations(True)
    assert o.dtype == torch.float32 and o.shape == o_ref.shape
    sig = o_ref.pow(2).mean().sqrt().item()
    r = (o.cpu().double() - o_ref.double()).pow(2).mean().sqrt().item()
    r32 = (o_f32act.cpu().double() - o_ref.double()).pow(2).mean().sqrt().item()
    record_stats("bf16wave", "generator_bf16act/" + widths, rms_err=r, rms_err_f32_activations=r32, signal_rms=sig)
    assert sig >= 0.1
    assert r <= 1e-3, "waveform RMS error %.3e with bf16 activations (%.3e with fp32 activations; signal RMS %.3e)" % (r, r32, sig)
