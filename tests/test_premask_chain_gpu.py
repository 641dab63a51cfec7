"""The discriminator step's pre-masked gradient chain (ops.premask_chain, model/discriminators/_pair.py): a convolution
whose input is another convolution's leaky-ReLU output applies that derivative in the epilogue of its own data-gradient
launch, and the producer skips its activation-derivative pass.  Same gradients as the separate pass (switch off), for the
multi-period and the multi-scale discriminator at base widths, fp32 and bf16 mode."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads(gpu, ops, cls, kw, on, seed=5, B=3, T=8192):
    from vcvits_amd.losses import discriminator_loss
    torch.manual_seed(seed)
    net = cls(**kw).to(gpu)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    y = torch.randn(B, 1, T, generator=g).to(gpu) * 0.3
    y_hat = torch.randn(B, 1, T, generator=g).to(gpu) * 0.3
    was = ops._PREMASK[0]
    ops._PREMASK[0] = on
    try:
        ops.invalidate_weights()
        before = ops.LAUNCH_COUNTS.get("premasked", 0)
        r, f, fr, ff = net(y, y_hat)
        loss, _, _ = discriminator_loss(r, f)
        loss.backward()
        fused = ops.LAUNCH_COUNTS.get("premasked", 0) - before
    finally:
        ops._PREMASK[0] = was
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.clone() for n, p in net.named_parameters()}, fused, fr


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("which", ["period", "scale"])
def test_premasked_chain_equals_separate_activation_pass(gpu, which, mode):
    from vcvits_amd import ops
    from vcvits_amd.model.discriminators.multi_period_discriminator import MultiPeriodDiscriminator
    from vcvits_amd.model.discriminators.multi_scale_discriminator import MultiScaleDiscriminator
    cls, kw = (MultiPeriodDiscriminator, dict(periods=[2, 3, 5])) if which == "period" else (MultiScaleDiscriminator, {})
    ops.set_compute_dtype(mode)
    try:
        # (scale: 32,768 samples, so that the first scale's ungrouped 1024 -> 1024 layer sees 128 frames -- at <= 64 frames
        # it runs in the folded batch layout, which keeps the separate pass)
        T, B = (8192, 3) if which == "period" else (32768, 2)
        l0, g0, n0, _ = _grads(gpu, ops, cls, kw, False, B=B, T=T)
        l1, g1, n1, fr = _grads(gpu, ops, cls, kw, True, B=B, T=T)
    finally:
        ops.set_compute_dtype("f32")
    assert n0 == 0
    # period: conv1..conv4 and conv_post of each DiscriminatorP consume a leaky conv output (the DiscriminatorS inside MPD sees
    # 32 frames at its ungrouped layer: folded layout, separate pass); scale: the ungrouped layer and conv_post of the first scale
    assert n1 == (3 * 5 if which == "period" else 2), n1
    assert l0 == l1
    assert all(not f.requires_grad for fm in fr for f in fm), "feature maps of the discriminator step must be recorded detached"
    for n in g0:
        a, b = g0[n].double(), g1[n].double()
        scale = float(a.abs().max()) + 1e-30
        err = float((a - b).abs().max()) / scale
        # identical products; only the bias sums are collected in another order (weight-gradient launch vs the fused pass)
        assert err <= (2e-6 if n.endswith("bias") else 1e-6), (n, err)
