"""tests/golden/disc_spectral.npz: the reference's discriminators built with use_spectral_norm=True
(vits/model/discriminators/discriminator.py:17,52 -> torch.nn.utils.spectral_norm), run in this container, and the oracle's
restatement checked against them.  Same conventions as tools/make_goldens.py (seeded state, checksums at full width).

Run:  python tools/make_goldens_spectral.py        (only here; /root/reference does not exist on the GPU box)

Captured: DiscriminatorS full width -- two consecutive TRAINING forwards (each advances the power-iteration vectors), the
vectors afterwards, then an eval forward; DiscriminatorP(3) training forward + gradients of a probe loss (input, weight_orig
of the first / last conv, a bias); MultiPeriodDiscriminator([2, 3]) and MultiScaleDiscriminator logits in training mode
(d(y) then d(y_hat): two power iterations per layer per call)."""
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

# (importing make_goldens installs the import stubs and puts /root/reference on the path)
from make_goldens import (DiscriminatorP, DiscriminatorS, MultiPeriodDiscriminator, MultiScaleDiscriminator, O, checksum,  # noqa: E402
                          close, grads_of, load_seeded, rng_tensor, save)


def sums(tag, outs):
    arrs = {}
    for i, t in enumerate(outs):
        s, idx, vals = checksum(t, seed=i)
        arrs["%s_sum_%d" % (tag, i)] = s
        arrs["%s_idx_%d" % (tag, i)] = idx
        arrs["%s_val_%d" % (tag, i)] = vals
        arrs["%s_shape_%d" % (tag, i)] = np.array(t.shape)
    return arrs


def osd(module, prefix):
    return {prefix + "." + k: v.clone() for k, v in module.state_dict().items()}


def main():
    rng = np.random.default_rng(20241004)
    arrs = {}
    # ---- DiscriminatorS, full width ----------------------------------------------------------------------------------
    B, TW = 2, 8192
    y1, y2 = rng_tensor(rng, (B, 1, TW), 0.3), rng_tensor(rng, (B, 1, TW), 0.3)
    ds = DiscriminatorS(use_spectral_norm=True)
    load_seeded(ds, 210)
    assert sorted(k.rsplit(".", 1)[-1] for k in ds.convs[0].state_dict()) == ["bias", "weight_orig", "weight_u", "weight_v"]
    sdo = osd(ds, "d")
    ds.train()
    with torch.no_grad():
        for k, yy in (("s_tr1", y1), ("s_tr2", y2)):
            logit, fmap = ds(yy)
            lo, fo = O.disc_s_forward(sdo, "d", yy, training=True)
            close(lo, logit, what="spectral DiscS logits " + k)
            for a_, b_ in zip(fo, fmap):
                close(a_, b_, what="spectral DiscS fmap " + k)
            arrs.update(sums(k, [logit] + fmap))
        for k, v in ds.state_dict().items():
            if k.endswith("weight_u") or k.endswith("weight_v"):
                close(sdo["d." + k], v, what="vector " + k)
                arrs["s_vec_" + k] = v
        ds.eval()
        logit, fmap = ds(y1)
        lo, fo = O.disc_s_forward(sdo, "d", y1, training=False)
        close(lo, logit, what="spectral DiscS eval")
        arrs.update(sums("s_ev", [logit] + fmap))
    # ---- DiscriminatorP(3): gradients ----------------------------------------------------------------------------------
    dp = DiscriminatorP(3, use_spectral_norm=True)
    load_seeded(dp, 211)
    dp.train()
    sdo = osd(dp, "d")
    ysm = rng_tensor(rng, (1, 1, 500), 0.3).requires_grad_(True)
    logit, fmap = dp(ysm)
    r = rng_tensor(rng, logit.shape)
    names = ["convs.0.weight_orig", "convs.0.bias", "convs.2.weight_orig", "convs.4.weight_orig", "conv_post.weight_orig",
             "conv_post.bias"]
    pd = dict(dp.named_parameters())
    gr = grads_of([logit, fmap[2]], [r, torch.ones_like(fmap[2]) * 0.01], [ysm] + [pd[n] for n in names])
    # the oracle's gradients (autograd through its restatement) against the reference's
    yo = ysm.detach().clone().requires_grad_(True)
    for k in list(sdo):
        sdo[k] = sdo[k].detach().clone().requires_grad_(k.endswith("weight_orig") or k.endswith("bias"))
    lo, fo = O.disc_p_forward(sdo, "d", yo, 3, training=True)
    go = grads_of([lo, fo[2]], [r, torch.ones_like(fo[2]) * 0.01], [yo] + [sdo["d." + n] for n in names])
    close(lo, logit, what="spectral DiscP logit")
    for n, a_, b_ in zip(["dy"] + names, go, gr):
        close(a_, b_, tol=2e-5, what="spectral DiscP grad " + n)
    arrs.update(dict(p_seed=211, p_y=ysm, p_r=r, p_logit=logit, p_dy=gr[0]))
    for n, gg in zip(names, gr[1:]):
        if gg.numel() <= 4096:
            arrs["p_dp_" + n] = gg
        else:
            arrs.update(sums("p_dps_" + n, [gg]))
    # ---- MPD / MSD in training mode ----------------------------------------------------------------------------------
    mpd = MultiPeriodDiscriminator(periods=[2, 3], use_spectral_norm=True)
    load_seeded(mpd, 212)
    msd = MultiScaleDiscriminator(use_spectral_norm=True)
    load_seeded(msd, 213)
    mpd.train(); msd.train()
    ya, yb = rng_tensor(rng, (1, 1, 2048), 0.3), rng_tensor(rng, (1, 1, 2048), 0.3)
    with torch.no_grad():
        so = osd(mpd, "m")
        r_, g_, _, _ = mpd(ya, yb)
        ro, go_, _, _ = O.mpd_forward(so, "m", ya, yb, [2, 3], training=True)
        for a_, b_ in zip(ro + go_, r_ + g_):
            close(a_, b_, what="spectral MPD")
        so = osd(msd, "m")
        rs_, gs_, _, _ = msd(ya, yb)
        ro, go_, _, _ = O.msd_forward(so, "m", ya, yb, training=True)
        for a_, b_ in zip(ro + go_, rs_ + gs_):
            close(a_, b_, what="spectral MSD")
    for i, t in enumerate(r_ + g_):
        arrs["mpd_%d" % i] = t
    for i, t in enumerate(rs_ + gs_):
        arrs["msd_%d" % i] = t
    save("disc_spectral.npz", seed_s=210, y1=y1, y2=y2, seed_mpd=212, seed_msd=213, ya=ya, yb=yb, **arrs)


if __name__ == "__main__":
    main()
