"""GPU: configs/48k_base.json shapes on the HIP path against the oracle -- the 12-period MPD (periods 13, 19, 29, 31
never occur in base.json), 128-wide posterior encoder / flow / content encoder, and the config-5 inference path at its
real length (938 frames = 10 s).  Oracle-vs-HIP on seeded inputs (NumPy-seeded weights, golden_util.fill_state_dict)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import close_kinked, fill_state_dict, keys_shapes_of

pytestmark = pytest.mark.gpu


def close(name, a, b, tol=1e-4, elem=True):
    """max-norm: |a-b| <= tol * max|b|; and element-wise: |a_i-b_i| <= 10 tol |b_i| + 0.2 tol max|b| (small elements
    must be right to a fifth of the max-norm budget, not just the large ones)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    mx = b.abs().max().item()
    err = (a - b).abs()
    assert err.max().item() <= tol * mx + 2e-6, "%s: max err %.3e vs %.3e" % (name, err.max().item(), tol * mx)
    if elem:
        bad = err > (10 * tol * b.abs() + 0.2 * tol * mx + 2e-6)
        assert not bad.any(), "%s: %d elements outside the element-wise bound" % (name, int(bad.sum()))


def pre(sd, p):
    return {p + "." + k: v for k, v in sd.items()}


@pytest.mark.parametrize("period", [13, 19, 29, 31])
def test_discriminator_p_48k_periods(gpu, period):
    from oracle import vits_oracle as O
    from vcvits_amd.model.discriminators.discriminator import DiscriminatorP
    d = DiscriminatorP(period)
    sd = fill_state_dict(keys_shapes_of(d), 100 + period)
    d.load_state_dict(sd)
    d = d.to(gpu)
    rng = np.random.default_rng(period)
    y = torch.from_numpy(rng.uniform(-0.9, 0.9, size=(2, 1, 16384)).astype(np.float32))
    yg = y.to(gpu).requires_grad_(True)
    logit, fmap = d(yg)
    yc = y.clone().requires_grad_(True)
    sdc = {k: v.clone().requires_grad_(True) for k, v in pre(sd, "d").items()}
    logit_o, fmap_o = O.disc_p_forward(sdc, "d", yc, period)
    close("logit p%d" % period, logit, logit_o)
    for i, (f, fo) in enumerate(zip(fmap, fmap_o)):
        close("fmap%d p%d" % (i, period), f, fo)
    r = torch.from_numpy(rng.standard_normal(tuple(logit_o.shape)).astype(np.float32))
    ((logit * r.to(gpu)).sum() + 1e-3 * fmap[3].sum()).backward()
    ((logit_o * r).sum() + 1e-3 * fmap_o[3].sum()).backward()
    close_kinked("dy p%d" % period, yg.grad, yc.grad)
    floor = 2e-6 * max(float(v.grad.abs().max()) for v in sdc.values())
    for n, p in d.named_parameters():
        close_kinked("d%s p%d" % (n, period), p.grad, sdc["d." + n].grad, floor=floor)


@pytest.mark.parametrize("period", [13, 19, 29, 31])
def test_period_conv_layers_strict(gpu, period):
    """Every conv layer of DiscriminatorP at the 48k-only periods as a LINEAR launch (no activation, so no kink):
    forward, data gradient and weight gradient against torch CPU, strict max-norm."""
    from vcvits_amd import ops
    chans = [1, 32, 128, 512, 1024, 1024]
    h = (16384 + period - 1) // period
    rng = np.random.default_rng(1000 + period)
    for i in range(5):
        s = 3 if i < 4 else 1
        ci, co = chans[i], chans[i + 1]
        x = torch.from_numpy(rng.standard_normal((2, ci, h, period)).astype(np.float32))
        w = torch.from_numpy((rng.standard_normal((co, ci, 5)) * (ci * 5) ** -0.5).astype(np.float32))
        b = torch.from_numpy((rng.standard_normal(co) * 0.1).astype(np.float32))
        xc, wc, bc = (t.clone().requires_grad_(True) for t in (x, w, b))
        y = F.conv2d(xc, wc.unsqueeze(-1), bc, stride=(s, 1), padding=(2, 0))
        r = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        (y * r).sum().backward()
        xg, wg, bg = (t.to(gpu).requires_grad_(True) for t in (x, w, b))
        yg = ops.conv1d(xg, wg.unsqueeze(-1), bg, stride=s, pad=2)
        (yg * r.to(gpu)).sum().backward()
        tag = "p%d layer %d" % (period, i)
        close(tag + " y", yg, y, tol=2e-5, elem=False)
        close(tag + " dx", xg.grad, xc.grad, tol=2e-5, elem=False)
        close(tag + " dw", wg.grad.reshape(wc.shape), wc.grad, tol=2e-5, elem=False)
        close(tag + " db", bg.grad, bc.grad, tol=2e-5, elem=False)
        assert float(wg.grad.abs().max()) > 0
        h = y.shape[2]


def test_posterior_flow_content_at_128(gpu):
    """48k widths: C = hidden = 128, heads 4 (d_k 32), hubert 768, T_y = 96 / T_x = 52 frames."""
    from oracle import vits_oracle as O
    from vcvits_amd.model.encoders.content_encoder import HubertContentEncoder
    from vcvits_amd.model.encoders.posterior_encoder import PosteriorEncoder
    from vcvits_amd.model.flow import ResidualCouplingBlock
    C, H, GIN, Ty, Tx = 128, 128, 256, 96, 52
    rng = np.random.default_rng(48)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    g_in = t(2, GIN, 1)
    ylen, xlen = torch.tensor([96, 70]), torch.tensor([52, 40])
    # posterior encoder
    enc = PosteriorEncoder(1025, C, H, 5, 1, 16, gin_channels=GIN)
    sd = fill_state_dict(keys_shapes_of(enc), 481)
    enc.load_state_dict(sd)
    enc = enc.to(gpu)
    spec, eps = t(2, 1025, Ty).abs(), t(2, C, Ty)
    sg, gg = spec.to(gpu).requires_grad_(True), g_in.to(gpu).requires_grad_(True)
    z, m, logs, mask = enc(sg, ylen.to(gpu), g=gg, noise=eps.to(gpu))
    sc, gc = spec.clone().requires_grad_(True), g_in.clone().requires_grad_(True)
    sdc = {k: v.clone().requires_grad_(True) for k, v in pre(sd, "e").items()}
    zo, mo, lo, masko = O.posterior_encoder_forward(sdc, "e", sc, ylen, gc, eps, C, H, 5, 1, 16)
    close("z", z, zo); close("m", m, mo); close("logs", logs, lo)
    rz = t(2, C, Ty)
    ((z * rz.to(gpu)).sum() + m.sum() * 0.1).backward()
    ((zo * rz).sum() + mo.sum() * 0.1).backward()
    close("dspec", sg.grad, sc.grad, tol=2e-4); close("dg", gg.grad, gc.grad, tol=2e-4)
    for n, p in enc.named_parameters():
        close("enc_q d" + n, p.grad, sdc["e." + n].grad, tol=3e-4, elem=False)
    # flow forward + reverse
    flow = ResidualCouplingBlock(C, H, 5, 1, 4, gin_channels=GIN)
    sdf = fill_state_dict(keys_shapes_of(flow), 482)
    flow.load_state_dict(sdf)
    flow = flow.to(gpu)
    zin = zo.detach()
    zg = zin.to(gpu).requires_grad_(True)
    zp = flow(zg, mask, g=gg.detach())
    sdfc = {k: v.clone().requires_grad_(True) for k, v in pre(sdf, "f").items()}
    zc = zin.clone().requires_grad_(True)
    zpo = O.flow_forward(sdfc, "f", zc, masko, g_in, False, C, H, 5, 1, 4)
    close("z_p", zp, zpo)
    (zp * rz.to(gpu)).sum().backward()
    (zpo * rz).sum().backward()
    close("dz", zg.grad, zc.grad, tol=2e-4)
    for n, p in flow.named_parameters():
        close("flow d" + n, p.grad, sdfc["f." + n].grad, tol=3e-4, elem=False)
    with torch.no_grad():
        zr = flow(zp.detach(), mask, g=gg.detach(), reverse=True)
    close("z_rev", zr, O.flow_forward(sdfc, "f", zpo.detach(), masko, g_in, True, C, H, 5, 1, 4).detach())
    close("roundtrip", zr, (zin * masko), tol=2e-5)
    # content encoder (feature input)
    ce = HubertContentEncoder(None, C, H, 768, 4, 3, 3, 0.1, 768, 512).eval()
    sde = fill_state_dict(keys_shapes_of(ce), 483)
    ce.load_state_dict(sde)
    ce = ce.to(gpu)
    feats = t(2, 768, Tx)
    pitch = torch.from_numpy(rng.integers(1, 512, size=(2, Tx)))
    fg = feats.to(gpu).requires_grad_(True)
    x, mp, lp, xm = ce(fg, xlen.to(gpu), pitch.to(gpu), xlen.to(gpu))
    sdec = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in pre(sde, "c").items()}
    fc = feats.clone().requires_grad_(True)
    xo, mpo, lpo, xmo = O.content_encoder_forward(sdec, "c", fc, xlen, pitch, C, 4, 3, 3)
    close("x", x, xo, tol=2e-4); close("m_p", mp, mpo, tol=2e-4); close("logs_p", lp, lpo, tol=2e-4)
    rx = t(2, C, Tx)
    ((mp * rx.to(gpu)).sum() + lp.sum() * 0.1).backward()
    ((mpo * rx).sum() + lpo.sum() * 0.1).backward()
    close_kinked("dfeats", fg.grad, fc.grad)  # the FFN's ReLU has a kink too
    floor = 2e-6 * max(float(v.grad.abs().max()) for v in sdec.values() if v.requires_grad)
    for n, p in ce.named_parameters():
        close_kinked("enc_p d" + n, p.grad, sdec["c." + n].grad, floor=floor)


def test_infer_48k_full_length(gpu):
    """BASELINE configs[4] at its real size, one utterance: 48k widths, 938 frames (10 s) -> 480256 samples; prior
    sample + flow reverse + decoder against the oracle composition."""
    from oracle import vits_oracle as O
    from vcvits_amd import configs, ops
    from vcvits_amd.model.synthesizers.synthesizer_svc import SynthesizerSVC
    cfg = configs.base_48k()
    d, m = cfg["data"], cfg["model"]
    net = SynthesizerSVC(d["filter_length"] // 2 + 1, 32, n_speakers=d["n_speakers"], **m)
    sd = fill_state_dict(keys_shapes_of(net), 485)
    net.load_state_dict(sd)
    net = net.to(gpu).eval()
    C, H, T = m["inter_channels"], m["hidden_channels"], 938
    rng = np.random.default_rng(5)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    m_p, logs_p, noise = t(1, C, T), t(1, C, T) * 0.1 - 1.0, t(1, C, T)
    sid = torch.tensor([17])
    mask = torch.ones(1, 1, T)
    with torch.no_grad():
        spk = net.emb_g(sid.to(gpu)).unsqueeze(-1)
        z_p = ops.prior_sample(m_p.to(gpu), logs_p.to(gpu), noise.to(gpu), 0.667)
        z = net.flow(z_p, mask.to(gpu), g=spk, reverse=True)
        o = net.dec(ops.mask_mul(z, mask.to(gpu).reshape(1, -1)))
        sdp = pre(sd, "n")
        g = F.embedding(sid, sd["emb_g.weight"]).unsqueeze(-1)
        zp_o = m_p + noise * torch.exp(logs_p) * 0.667
        z_o = O.flow_forward(sdp, "n.flow", zp_o, mask, g, True, C, H, 5, 1, 4)
        o_o = O.generator_forward(sdp, "n.dec", z_o * mask, m["upsample_rates"], m["upsample_kernel_sizes"],
                                  m["resblock_kernel_sizes"], m["resblock_dilation_sizes"])
    assert o.shape == (1, 1, 938 * 512)
    close("z_p", z_p, zp_o, tol=1e-5); close("z", z, z_o)
    close("o", o, o_o, tol=2e-4)
    rms = ((o.cpu().double() - o_o.double()) ** 2).mean().sqrt().item()
    assert rms <= 1e-5, rms
