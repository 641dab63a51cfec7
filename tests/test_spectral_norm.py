"""use_spectral_norm=True (reference: vits/model/discriminators/discriminator.py:17,52 -> torch.nn.utils.spectral_norm).

CPU: the oracle's restatement against the vectors captured from the reference's own discriminators
(tools/make_goldens_spectral.py) and against torch.nn.utils.spectral_norm itself; the product modules' state_dict surface.
GPU (-m gpu): the HIP path (vcv_spectral_norm_fwd / _bwd behind ops.spectral_norm) against the same vectors: two training
forwards (the power-iteration vectors advance in place), an eval forward, gradients, MPD / MSD in training mode."""
import numpy as np
import pytest
import torch

from golden_util import checksum, fill_state_dict, keys_shapes_of, load
from oracle import vits_oracle as O


def T(x, dev=None, grad=False):
    t = torch.from_numpy(np.asarray(x))
    if dev is not None:
        t = t.to(dev)
    return t.requires_grad_(True) if grad else t


def close(name, a, b, tol=1e-4, atol=2e-6):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    bound = tol * b.abs().max().item() + atol
    assert err <= bound, "%s: abs err %.3e > %.3e" % (name, err, bound)


def check_sums(g, tag, outs, tol=5e-5):
    for i, t in enumerate(outs):
        assert tuple(g["%s_shape_%d" % (tag, i)]) == tuple(t.shape)
        s, idx, vals = checksum(t, seed=i)
        np.testing.assert_array_equal(idx, g["%s_idx_%d" % (tag, i)])
        ref_s = g["%s_sum_%d" % (tag, i)]
        assert abs(s[1] - ref_s[1]) <= tol * abs(ref_s[1]) + 1e-6, (tag, i)
        assert abs(s[0] - ref_s[0]) <= tol * abs(ref_s[1]) + 1e-6, (tag, i)
        scale = np.abs(g["%s_val_%d" % (tag, i)]).max() + 1e-9
        assert np.abs(vals - g["%s_val_%d" % (tag, i)]).max() <= 2 * tol * scale + 1e-6, (tag, i)


def sd_for(module, seed, prefix):
    sd = fill_state_dict(keys_shapes_of(module), int(seed))
    module.load_state_dict(sd)  # strict: key names and shapes equal the reference's (the fill is keyed by them)
    return {prefix + "." + k: v.clone() for k, v in sd.items()}


# ---- CPU ------------------------------------------------------------------------------------------------------------
def test_state_dict_surface_matches_torch_spectral_norm():
    from vcvits_amd.model.discriminators.discriminator import DiscriminatorP, DiscriminatorS
    from torch.nn.utils import spectral_norm
    ref = spectral_norm(torch.nn.Conv1d(16, 64, 41, 4, groups=4, padding=20))
    mine = DiscriminatorS(use_spectral_norm=True).convs[1]
    assert [(k, tuple(v.shape)) for k, v in ref.state_dict().items()] == keys_shapes_of(mine)
    assert [n for n, _ in ref.named_parameters()] == [n for n, _ in mine.named_parameters()] == ["bias", "weight_orig"]
    assert abs(float(mine.weight_u.norm()) - 1) < 1e-5 and abs(float(mine.weight_v.norm()) - 1) < 1e-5
    ref2 = spectral_norm(torch.nn.Conv2d(32, 128, (5, 1), (3, 1), padding=(2, 0)))
    mine2 = DiscriminatorP(3, use_spectral_norm=True).convs[1]
    assert [(k, tuple(v.shape)) for k, v in ref2.state_dict().items()] == keys_shapes_of(mine2)
    # the multi-scale stack norms only its first discriminator spectrally (multi_scale_discriminator.py:13-19)
    from vcvits_amd.model.discriminators.multi_scale_discriminator import MultiScaleDiscriminator
    keys = [k for k, _ in keys_shapes_of(MultiScaleDiscriminator(use_spectral_norm=True))]
    assert "discriminators.0.convs.0.weight_orig" in keys and "discriminators.1.convs.0.weight_g" in keys


@pytest.mark.parametrize("training", [True, False])
def test_oracle_weight_equals_torch_spectral_norm(training):
    from torch.nn.utils import spectral_norm
    torch.manual_seed(3)
    conv = spectral_norm(torch.nn.Conv1d(8, 12, 5))
    conv.train(training)
    sd = {"c." + k: v.clone() for k, v in conv.state_dict().items()}
    x = torch.randn(2, 8, 20)
    for _ in range(2):  # (training: the vectors advance with every forward)
        y = conv(x)
        w = O.spectral_norm_weight(sd, "c", training)
        assert torch.allclose(torch.nn.functional.conv1d(x, w, sd["c.bias"]), y, rtol=1e-5, atol=1e-6)
    assert torch.allclose(sd["c.weight_u"], conv.weight_u, atol=1e-6) and torch.allclose(sd["c.weight_v"], conv.weight_v, atol=1e-6)


def test_oracle_vs_reference_vectors():
    from vcvits_amd.model.discriminators.discriminator import DiscriminatorP, DiscriminatorS
    g = load("disc_spectral.npz")
    sd = sd_for(DiscriminatorS(use_spectral_norm=True), g["seed_s"], "d")
    with torch.no_grad():
        for tag, y in (("s_tr1", T(g["y1"])), ("s_tr2", T(g["y2"]))):
            logit, fmap = O.disc_s_forward(sd, "d", y, training=True)
            check_sums(g, tag, [logit] + fmap, tol=2e-5)
        for k in g.files:
            if k.startswith("s_vec_"):
                close(k, sd["d." + k[6:]], g[k], tol=2e-5)
        logit, fmap = O.disc_s_forward(sd, "d", T(g["y1"]), training=False)
        check_sums(g, "s_ev", [logit] + fmap, tol=2e-5)
    sd = sd_for(DiscriminatorP(3, use_spectral_norm=True), g["p_seed"], "d")
    logit, _ = O.disc_p_forward(sd, "d", T(g["p_y"]), 3, training=True)
    close("p_logit", logit, g["p_logit"], tol=2e-5)


# ---- GPU ------------------------------------------------------------------------------------------------------------
def build(module, seed, dev):
    module.load_state_dict(fill_state_dict(keys_shapes_of(module), int(seed)))
    return module.to(dev)


@pytest.mark.gpu
def test_kernel_against_torch_ops(gpu):
    """ops.spectral_norm alone: weight, in-place vector update and the weight gradient against the same arithmetic in
    torch on the device (fp32 reference of the op), training and eval, at the largest layer shape and a ragged one."""
    from vcvits_amd import ops
    for R, rest in ((1024, (1024, 5, 1)), (37, (3, 7)), (1, (1024, 3))):
        for training in (True, False):
            torch.manual_seed(R)
            w = (torch.randn((R,) + rest, device=gpu) * 0.05).requires_grad_(True)
            N = w[0].numel()
            u = torch.nn.functional.normalize(torch.randn(R, device=gpu), dim=0)
            v = torch.nn.functional.normalize(torch.randn(N, device=gpu), dim=0)
            u2, v2 = u.clone(), v.clone()
            wsn = ops.spectral_norm(w, u, v, training)
            r = torch.randn_like(wsn)
            (wsn * r).sum().backward()
            wr = w.detach().clone().double().requires_grad_(True)
            wm = wr.reshape(R, -1)
            u2, v2 = u2.double(), v2.double()
            if training:
                with torch.no_grad():
                    v2 = torch.nn.functional.normalize(wm.t() @ u2, dim=0, eps=1e-12)
                    u2 = torch.nn.functional.normalize(wm @ v2, dim=0, eps=1e-12)
            ref = wr / torch.dot(u2, wm @ v2)
            (ref * r.double()).sum().backward()
            close("w_sn", wsn, ref.detach().cpu().numpy(), tol=2e-6)
            close("u", u, u2.cpu().numpy(), tol=2e-6)
            close("v", v, v2.cpu().numpy(), tol=2e-6)
            close("dw", w.grad, wr.grad.cpu().numpy(), tol=5e-6)


@pytest.mark.gpu
def test_discriminator_s_full_width(gpu):
    from vcvits_amd.model.discriminators.discriminator import DiscriminatorS
    g = load("disc_spectral.npz")
    d = build(DiscriminatorS(use_spectral_norm=True), g["seed_s"], gpu)
    d.train()
    with torch.no_grad():
        for tag, y in (("s_tr1", T(g["y1"], gpu)), ("s_tr2", T(g["y2"], gpu))):
            logit, fmap = d(y)
            check_sums(g, tag, [logit] + fmap)
        sd = d.state_dict()
        for k in g.files:
            if k.startswith("s_vec_"):
                close(k, sd[k[6:]], g[k])
        d.eval()
        logit, fmap = d(T(g["y1"], gpu))
        check_sums(g, "s_ev", [logit] + fmap)
        # an eval forward leaves the vectors alone
        for k in g.files:
            if k.startswith("s_vec_"):
                close(k, d.state_dict()[k[6:]], g[k])


@pytest.mark.gpu
def test_discriminator_p_gradients(gpu):
    from vcvits_amd.model.discriminators.discriminator import DiscriminatorP
    g = load("disc_spectral.npz")
    d = build(DiscriminatorP(3, use_spectral_norm=True), g["p_seed"], gpu)
    d.train()
    y = T(g["p_y"], gpu, True)
    logit, fmap = d(y)
    close("logit", logit, g["p_logit"])
    ((logit * T(g["p_r"], gpu)).sum() + 0.01 * fmap[2].sum()).backward()
    close("dy", y.grad, g["p_dy"])
    seen = 0
    for n, p in d.named_parameters():
        if "p_dp_" + n in g.files:
            close(n, p.grad, g["p_dp_" + n])
            seen += 1
        elif "p_dps_%s_sum_0" % n in g.files:
            check_sums(g, "p_dps_" + n, [p.grad], tol=1e-4)
            seen += 1
    assert seen == 6


@pytest.mark.gpu
def test_mpd_msd_training_mode(gpu):
    """d(y) then d(y_hat) are two forwards of a spectrally normed discriminator (two power iterations): the stacked pass of
    the weight-normed path must not be taken; with weights taking gradients (D step) and frozen (G step)."""
    from vcvits_amd.model.discriminators.multi_period_discriminator import MultiPeriodDiscriminator
    from vcvits_amd.model.discriminators.multi_scale_discriminator import MultiScaleDiscriminator
    g = load("disc_spectral.npz")
    ya, yb = T(g["ya"], gpu), T(g["yb"], gpu)
    for frozen in (False, True):
        mpd = build(MultiPeriodDiscriminator(periods=[2, 3], use_spectral_norm=True), g["seed_mpd"], gpu)
        msd = build(MultiScaleDiscriminator(use_spectral_norm=True), g["seed_msd"], gpu)
        mpd.train(); msd.train()
        for p in list(mpd.parameters()) + list(msd.parameters()):
            p.requires_grad_(not frozen)
        yh = yb.clone().requires_grad_(frozen)
        r, gg, fr, fg = mpd(ya, yh)
        for i, t in enumerate(r + gg):
            close("mpd_%d" % i, t, g["mpd_%d" % i])
        r, gg, _, _ = msd(ya, yh)
        for i, t in enumerate(r + gg):
            close("msd_%d" % i, t, g["msd_%d" % i])
        sum(t.sum() for t in gg).backward()  # both passes' graphs are intact
        if frozen:
            assert yh.grad is not None and torch.isfinite(yh.grad).all()
