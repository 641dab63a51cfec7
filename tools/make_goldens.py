"""Generate tests/golden/*.npz by importing the reference's importable modules
(/root/reference, read-only) in the authoring container, and check the oracle against them.

Run:  python tools/make_goldens.py         (only here; /root/reference does not exist on the GPU box)

What is captured: reduced-width instances of every hot-path module the
reference can construct -- inputs, parameters (as a seed for tests/golden_util.fill_state_dict),
outputs, and input/parameter gradients of the probe loss sum(out * r); full-width checksums for
the fixed-width discriminators; STFT/mel vectors; the four losses; the commons helpers.
Fixtures are DATA (inputs and expected outputs); no reference source is stored.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

torch.set_num_threads(8)

from golden_util import GOLDEN_DIR, checksum, fill_state_dict, keys_shapes_of, rng_tensor  # noqa: E402
from oracle import vits_oracle as O  # noqa: E402

REF = "/root/reference"


def _install_stubs():
    """librosa / torchaudio / fairseq are not installed; vits.mel_processing and
    content_encoder import them at module top.  Stub them (the mel filterbank stub is OUR
    Appendix-B restatement: it makes the import work, it does not pin librosa)."""
    lib = types.ModuleType("librosa")
    util = types.ModuleType("librosa.util")
    util.normalize = util.pad_center = util.tiny = lambda *a, **k: None
    filt = types.ModuleType("librosa.filters")
    filt.mel = lambda sr, n_fft, n_mels, fmin, fmax: O.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    lib.util, lib.filters = util, filt
    sys.modules.update({"librosa": lib, "librosa.util": util, "librosa.filters": filt})
    ta = types.ModuleType("torchaudio")
    taf = types.ModuleType("torchaudio.functional")
    tat = types.ModuleType("torchaudio.transforms")
    ta.functional, ta.transforms = taf, tat
    sys.modules.update({"torchaudio": ta, "torchaudio.functional": taf, "torchaudio.transforms": tat})

    class _FakeHubert(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.feats = None

        def extract_features(self, wav):
            return self.feats, None

    fs = types.ModuleType("fairseq")
    cu = types.ModuleType("fairseq.checkpoint_utils")
    cu.load_model_ensemble_and_task = lambda paths: ([_FakeHubert()], None, None)
    fs.checkpoint_utils = cu
    sys.modules.update({"fairseq": fs, "fairseq.checkpoint_utils": cu})


_install_stubs()
sys.path.insert(0, REF)

import vits.commons as rcommons  # noqa: E402
import vits.light.losses as rlosses  # noqa: E402
import vits.mel_processing as rmel  # noqa: E402
import vits.model.modules as rmodules  # noqa: E402
from vits.model.discriminators.discriminator import DiscriminatorP, DiscriminatorS  # noqa: E402
from vits.model.discriminators.multi_period_discriminator import MultiPeriodDiscriminator  # noqa: E402
from vits.model.discriminators.multi_scale_discriminator import MultiScaleDiscriminator  # noqa: E402
from vits.model.encoders.content_encoder import HubertContentEncoder, PreloadHubertContentEncoder  # noqa: E402
from vits.model.encoders.posterior_encoder import PosteriorEncoder  # noqa: E402
from vits.model.flow import ResidualCouplingBlock  # noqa: E402
from vits.model.transformer.relative_attention_transformer import (MultiHeadAttention,  # noqa: E402
                                                                    TransformerEncoder)


def load_seeded(module, seed):
    sd = fill_state_dict(keys_shapes_of(module), seed)
    module.load_state_dict(sd)
    return sd


def close(a, b, tol=1e-5, what=""):
    a, b = a.detach(), b.detach()
    err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
    assert err < tol, "oracle mismatch %s: %.3e" % (what, err)
    return err


def grads_of(outputs, probes, leaves):
    loss = sum((o * r).sum() for o, r in zip(outputs, probes))
    return torch.autograd.grad(loss, leaves, allow_unused=True)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(GOLDEN_DIR, name), **out)
    print("wrote", name, sum(a.nbytes for a in out.values()) // 1024, "KiB")


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    rng = np.random.default_rng(20240601)

    # ---- WN ----------------------------------------------------------------------------------
    H, GIN, T, B = 16, 8, 24, 2
    wn = rmodules.WN(H, 5, 1, 3, gin_channels=GIN)
    seed = 101
    sd = load_seeded(wn, seed)
    x = rng_tensor(rng, (B, H, T)).requires_grad_(True)
    g = rng_tensor(rng, (B, GIN, 1)).requires_grad_(True)
    lengths = torch.tensor([24, 17])
    mask = O.sequence_mask(lengths, T).unsqueeze(1).float()
    y = wn(x, mask, g=g)
    r = rng_tensor(rng, y.shape)
    params = [p for _, p in wn.named_parameters()]
    gr = grads_of([y], [r], [x, g] + params)
    yo = O.wn_forward({k: v for k, v in wn.state_dict().items()}, "", x, mask, g, H, 5, 1, 3) \
        if False else O.wn_forward({"w." + k: v for k, v in wn.state_dict().items()}, "w", x, mask, g, H, 5, 1, 3)
    close(yo, y, what="WN")
    save("wn.npz", seed=seed, x=x, g=g, lengths=lengths, y=y, r=r, dx=gr[0], dg=gr[1],
         **{"dp_" + n: gg for (n, _), gg in zip(wn.named_parameters(), gr[2:])})

    # ---- fused gate ----------------------------------------------------------------------------
    a = rng_tensor(rng, (2, 2 * H, 9))
    bb = rng_tensor(rng, (2, 2 * H, 1))
    acts = rcommons.fused_add_tanh_sigmoid_multiply(a, bb, torch.IntTensor([H]))
    save("gate.npz", a=a, b=bb, acts=acts)

    # ---- PosteriorEncoder ----------------------------------------------------------------------
    IN, OUT = 33, 8
    pe = PosteriorEncoder(IN, OUT, H, 5, 1, 3, gin_channels=GIN)
    seed = 102
    load_seeded(pe, seed)
    spec = rng_tensor(rng, (B, IN, T)).abs().requires_grad_(True)
    g = rng_tensor(rng, (B, GIN, 1)).requires_grad_(True)
    eps = rng_tensor(rng, (B, OUT, T))
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: eps
    try:
        z, m, logs, xm = pe(spec, lengths, g=g)
    finally:
        torch.randn_like = orig
    sdo = {"e." + k: v for k, v in pe.state_dict().items()}
    zo, mo, lo, xmo = O.posterior_encoder_forward(sdo, "e", spec, lengths, g, eps, OUT, H, 5, 1, 3)
    close(zo, z, what="posterior z"); close(mo, m, what="posterior m"); close(lo, logs, what="posterior logs")
    rz, rm, rl = (rng_tensor(rng, z.shape) for _ in range(3))
    params = [p for _, p in pe.named_parameters()]
    gr = grads_of([z, m, logs], [rz, rm, rl], [spec, g] + params)
    save("posterior.npz", seed=seed, spec=spec, g=g, eps=eps, lengths=lengths, z=z, m=m, logs=logs, mask=xm,
         rz=rz, rm=rm, rl=rl, dspec=gr[0], dg=gr[1],
         **{"dp_" + n: gg for (n, _), gg in zip(pe.named_parameters(), gr[2:])})

    # ---- flow forward / reverse ----------------------------------------------------------------
    C = 8
    fl = ResidualCouplingBlock(C, H, 5, 1, 2, n_flows=4, gin_channels=GIN)
    seed = 103
    load_seeded(fl, seed)  # randomises the zero-initialised `post` convs too
    zin = rng_tensor(rng, (B, C, T)).requires_grad_(True)
    g = rng_tensor(rng, (B, GIN, 1)).requires_grad_(True)
    zp = fl(zin, mask, g=g)
    zrev = fl(zp.detach(), mask, g=g, reverse=True)
    sdo = {"f." + k: v for k, v in fl.state_dict().items()}
    close(O.flow_forward(sdo, "f", zin, mask, g, False, C, H, 5, 1, 2), zp, what="flow fwd")
    close(O.flow_forward(sdo, "f", zp.detach(), mask, g, True, C, H, 5, 1, 2), zrev, what="flow rev")
    r = rng_tensor(rng, zp.shape)
    params = [p for _, p in fl.named_parameters()]
    gr = grads_of([zp], [r], [zin, g] + params)
    save("flow.npz", seed=seed, z=zin, g=g, lengths=lengths, z_p=zp, z_rev=zrev, r=r, dz=gr[0], dg=gr[1],
         **{"dp_" + n: gg for (n, _), gg in zip(fl.named_parameters(), gr[2:])})

    # ---- attention + transformer encoder ---------------------------------------------------------
    HC, FC, NH, NL, TT = 16, 48, 2, 2, 30
    mha = MultiHeadAttention(HC, HC, NH, p_dropout=0.0, window_size=4).eval()
    seed = 104
    load_seeded(mha, seed)
    xa = rng_tensor(rng, (B, HC, TT)).requires_grad_(True)
    la = torch.tensor([30, 21])
    xm = O.sequence_mask(la, TT).unsqueeze(1).float()
    am = xm.unsqueeze(2) * xm.unsqueeze(-1)
    ya = mha(xa, xa, attn_mask=am)
    sdo = {"a." + k: v for k, v in mha.state_dict().items()}
    yo, po = O.rel_attention(sdo, "a", xa, am, NH, 4)
    close(yo, ya, what="attention out"); close(po, mha.attn, what="attention probs")
    r = rng_tensor(rng, ya.shape)
    params = [p for _, p in mha.named_parameters()]
    gr = grads_of([ya], [r], [xa] + params)
    save("attention.npz", seed=seed, x=xa, lengths=la, y=ya, attn=mha.attn, r=r, dx=gr[0],
         **{"dp_" + n: gg for (n, _), gg in zip(mha.named_parameters(), gr[1:])})

    enc = TransformerEncoder(HC, FC, NH, NL, kernel_size=3, p_dropout=0.0, window_size=4).eval()
    seed = 105
    load_seeded(enc, seed)
    xe = rng_tensor(rng, (B, HC, TT)).requires_grad_(True)
    ye = enc(xe, xm)
    sdo = {"t." + k: v for k, v in enc.state_dict().items()}
    close(O.transformer_encoder_forward(sdo, "t", xe, xm, NH, NL, 3), ye, what="transformer")
    r = rng_tensor(rng, ye.shape)
    params = [p for _, p in enc.named_parameters()]
    gr = grads_of([ye], [r], [xe] + params)
    save("transformer.npz", seed=seed, x=xe, lengths=la, y=ye, r=r, dx=gr[0],
         **{"dp_" + n: gg for (n, _), gg in zip(enc.named_parameters(), gr[1:])})

    # ---- content encoders (post-HuBERT part; HuBERT itself is stubbed: out of scope) -------------
    HUB, NP = 20, 32
    for name, cls, preload, seed in (("content_hubert.npz", HubertContentEncoder, False, 106),
                                     ("content_preload.npz", PreloadHubertContentEncoder, True, 107)):
        if preload:
            ce = cls(C, HC, FC, NH, NL, 3, 0.0, HUB, NP).eval()
        else:
            ce = cls("stub.pt", C, HC, FC, NH, NL, 3, 0.0, HUB, NP).eval()
        ks = [(k, s) for k, s in keys_shapes_of(ce) if not k.startswith("hubert.")]
        sd = fill_state_dict(ks, seed)
        ce.load_state_dict(sd, strict=False)
        feats = rng_tensor(rng, (B, HUB, TT))
        pitch = torch.from_numpy(rng.integers(1, NP, size=(B, TT)))
        if preload:
            out = ce(feats, la, pitch, la)
        else:
            ce.hubert.feats = feats.transpose(1, 2)
            out = ce(torch.zeros(B, 1, 320 * TT), la, pitch, la)
        sdo = {"c." + k: v for k, v in sd.items()}
        oo = O.content_encoder_forward(sdo, "c", feats, la, pitch, C, NH, NL, 3, preload=preload)
        for a_, b_, w_ in zip(oo, out, ("x", "m", "logs", "mask")):
            close(a_, b_, what=name + " " + w_)
        save(name, seed=seed, feats=feats, pitch=pitch, lengths=la, x=out[0], m=out[1], logs=out[2],
             mask=out[3])

    # ---- ResBlock1 -------------------------------------------------------------------------------
    for k, seed in ((3, 108), (7, 109)):
        rb = rmodules.ResBlock1(8, k, (1, 3, 5))
        load_seeded(rb, seed)
        xr = rng_tensor(rng, (B, 8, 64)).requires_grad_(True)
        yr = rb(xr)
        sdo = {"r." + kk: v for kk, v in rb.state_dict().items()}
        close(O.resblock1_forward(sdo, "r", xr, k), yr, what="resblock1")
        r = rng_tensor(rng, yr.shape)
        params = [p for _, p in rb.named_parameters()]
        gr = grads_of([yr], [r], [xr] + params)
        save("resblock1_k%d.npz" % k, seed=seed, x=xr, y=yr, r=r, dx=gr[0],
             **{"dp_" + n: gg for (n, _), gg in zip(rb.named_parameters(), gr[1:])})

    # ---- discriminators: full width, checksums ---------------------------------------------------
    def disc_checks(tag, outs):
        arrs = {}
        for i, t in enumerate(outs):
            s, idx, vals = checksum(t, seed=i)
            arrs["%s_sum_%d" % (tag, i)] = s
            arrs["%s_idx_%d" % (tag, i)] = idx
            arrs["%s_val_%d" % (tag, i)] = vals
            arrs["%s_shape_%d" % (tag, i)] = np.array(t.shape)
        return arrs

    TW = 8192
    ywav = rng_tensor(rng, (B, 1, TW), 0.3)
    ds = DiscriminatorS()
    seed = 110
    load_seeded(ds, seed)
    with torch.no_grad():
        logit, fmap = ds(ywav)
        sdo = {"d." + k: v for k, v in ds.state_dict().items()}
        lo, fo = O.disc_s_forward(sdo, "d", ywav)
    close(lo, logit, what="DiscS logits")
    for a_, b_ in zip(fo, fmap):
        close(a_, b_, what="DiscS fmap")
    arrs = disc_checks("s", [logit] + fmap)
    for period, seed in ((2, 111), (3, 112), (37, 113)):
        dp = DiscriminatorP(period)
        load_seeded(dp, seed)
        Tp = 4099  # not divisible by 2, 3 or 37 -> reflect pad
        yp = ywav[:, :, :Tp]
        with torch.no_grad():
            logit, fmap = dp(yp)
            sdo = {"d." + k: v for k, v in dp.state_dict().items()}
            lo, fo = O.disc_p_forward(sdo, "d", yp, period)
        close(lo, logit, what="DiscP logits")
        for a_, b_ in zip(fo, fmap):
            close(a_, b_, what="DiscP fmap")
        arrs.update(disc_checks("p%d" % period, [logit] + fmap))
        arrs["p%d_seed" % period] = seed
    save("discriminators.npz", seed_s=110, y=ywav, tp=4099, **arrs)

    # small DiscP gradient golden (period 3; input + first/last conv grads)
    dp = DiscriminatorP(3)
    seed = 114
    load_seeded(dp, seed)
    ysm = rng_tensor(rng, (1, 1, 500), 0.3).requires_grad_(True)
    logit, fmap = dp(ysm)
    r = rng_tensor(rng, logit.shape)
    names = ["convs.0.weight_v", "convs.0.weight_g", "convs.0.bias", "convs.4.weight_g", "conv_post.weight_v",
             "conv_post.bias"]
    pd = dict(dp.named_parameters())
    gr = grads_of([logit, fmap[2]], [r, torch.ones_like(fmap[2]) * 0.01], [ysm] + [pd[n] for n in names])
    save("discp_grad.npz", seed=seed, y=ysm, r=r, logit=logit, dy=gr[0],
         **{"dp_" + n: gg for n, gg in zip(names, gr[1:])})

    # MPD / MSD structure (which discriminators, in which order, on which inputs): logits only
    mpd = MultiPeriodDiscriminator(periods=[2, 3])
    seed = 115
    load_seeded(mpd, seed)
    msd = MultiScaleDiscriminator()
    seed2 = 116
    load_seeded(msd, seed2)
    y1 = rng_tensor(rng, (1, 1, 2048), 0.3)
    y2 = rng_tensor(rng, (1, 1, 2048), 0.3)
    with torch.no_grad():
        r_, g_, fr_, fg_ = mpd(y1, y2)
        ro, go, fro, fgo = O.mpd_forward({"m." + k: v for k, v in mpd.state_dict().items()}, "m", y1, y2, [2, 3])
        for a_, b_ in zip(ro + go, r_ + g_):
            close(a_, b_, what="MPD")
        rs_, gs_, frs_, fgs_ = msd(y1, y2)
        ro, go, fro, fgo = O.msd_forward({"m." + k: v for k, v in msd.state_dict().items()}, "m", y1, y2)
        for a_, b_ in zip(ro + go, rs_ + gs_):
            close(a_, b_, what="MSD")
        fl_ref = rlosses.feature_loss(fr_, fg_)
        close(O.feature_loss(fr_, fg_), fl_ref, what="feature loss")
    arrs = {}
    for i, t in enumerate(r_ + g_):
        arrs["mpd_%d" % i] = t
    for i, t in enumerate(rs_ + gs_):
        arrs["msd_%d" % i] = t
    save("mpd_msd.npz", seed_mpd=115, seed_msd=116, y=y1, y_hat=y2, feature_loss=fl_ref, **arrs)

    # ---- losses ----------------------------------------------------------------------------------
    dr = [rng_tensor(rng, (2, n)) for n in (7, 30)]
    dg = [rng_tensor(rng, (2, n)) for n in (7, 30)]
    dl, _, _ = rlosses.discriminator_loss(dr, dg)
    gl, _ = rlosses.generator_loss(dg)
    close(O.discriminator_loss(dr, dg), dl, what="disc loss")
    close(O.generator_loss(dg), gl, what="gen loss")
    zp_, lq_, mp_, lp_ = (rng_tensor(rng, (2, 8, 24)) for _ in range(4))
    kl = rlosses.kl_loss(zp_, lq_, mp_, lp_, mask)
    close(O.kl_loss(zp_, lq_, mp_, lp_, mask), kl, what="kl")
    save("losses.npz", dr0=dr[0], dr1=dr[1], dg0=dg[0], dg1=dg[1], disc_loss=dl, gen_loss=gl, z_p=zp_,
         logs_q=lq_, m_p=mp_, logs_p=lp_, lengths=lengths, kl=kl)

    # ---- commons ----------------------------------------------------------------------------------
    xs = rng_tensor(rng, (3, 4, 40))
    ids = torch.tensor([0, 5, 28])
    seg = rcommons.slice_segments(xs, ids, 12)
    close(O.slice_segments(xs, ids, 12), seg, what="slice")
    lens = torch.tensor([40, 30, 13])
    torch.manual_seed(7)
    u = torch.rand([3])
    torch.manual_seed(7)
    seg2, ids2 = rcommons.rand_slice_segments(xs, lens, 12)
    assert torch.equal(O.slice_ids_from_uniform(u, lens, 12), ids2)
    sm = rcommons.sequence_mask(lens, 45)
    assert torch.equal(O.sequence_mask(lens, 45), sm)
    save("commons.npz", x=xs, ids=ids, seg=seg, lens=lens, u=u, ids_rand=ids2, seg_rand=seg2, seqmask=sm)

    # ---- STFT / mel --------------------------------------------------------------------------------
    n = np.arange(16384)
    sig = 0.5 * np.sin(2 * np.pi * 440.0 * n / 48000.0) + 0.2 * rng.standard_normal(16384)
    sig = np.clip(sig, -0.99, 0.99).astype(np.float32)
    ysig = torch.from_numpy(np.stack([sig, sig[::-1].copy()]))
    spec_reflect = rmel.spectrogram_torch(ysig, 2048, 48000, 512, 2048)
    close(O.spectrogram(ysig, 2048, 512, 2048, True), spec_reflect, what="spectrogram_torch")
    rmel.mel_basis.clear()
    mel_reflect = rmel.mel_spectrogram_torch(ysig, 2048, 128, 48000, 512, 2048, 0.0, None)
    melmat = torch.from_numpy(O.mel_filterbank(48000, 2048, 128, 0.0, None))
    close(O.spec_to_mel(spec_reflect, melmat), mel_reflect, what="mel_spectrogram_torch")
    rmel.mel_basis.clear()
    mel2 = rmel.spec_to_mel_torch(spec_reflect, 2048, 128, 48000, 0.0, None)
    close(mel2, mel_reflect, what="spec_to_mel_torch")
    # zero-pad variant: torch.stft is available here, torchaudio is not (Appendix C) -> vectors come
    # from the oracle restatement and are flagged as such in the file
    spec_zero = O.spectrogram(ysig, 2048, 512, 2048, False)
    save("stft_mel.npz", y=ysig, spec_reflect=spec_reflect, mel_reflect_128=mel_reflect,
         spec_zero_oracle=spec_zero, melmat_sum=np.array([melmat.double().sum().item(),
                                                          melmat.double().abs().max().item()]))
    print("all goldens written; oracle agrees with the reference imports")


if __name__ == "__main__":
    main()
