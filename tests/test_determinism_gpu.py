"""GPU: deterministic mode (ops.set_deterministic / VCVITS_DETERMINISTIC=1) makes the GAN training step of the vocoder
workload bit-reproducible: two identical steps from identical state give identical gradients for every parameter of
both optimizers -- including the launches the MFMA slab path does not cover (thin / grouped / register-staged weight
gradients, bias sums, activation-derivative bias sums), which run unsplit in this mode."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads(gpu, cfg, state, batch):
    from vcvits_amd.light.vcvits import VocoderGAN
    module = VocoderGAN(**cfg)
    module.load_state_dict(copy.deepcopy(state))
    module = module.to(gpu)
    module.configure_optimizers()
    out = {}

    def probe(idx, opt):
        out[idx] = opt.grad.detach().clone()

    losses = module.fit_batch({k: v.to(gpu) for k, v in batch.items()}, after_backward=probe)
    torch.cuda.synchronize()
    module.optim_g.close()
    module.optim_d.close()
    return out, losses


@pytest.mark.parametrize("widths", ["reduced", "base"])
def test_two_identical_steps_give_identical_gradients(gpu, widths):
    from vcvits_amd import configs, ops, synthetic
    from vcvits_amd.light.vcvits import VocoderGAN
    torch.manual_seed(0)
    cfg = configs.base()
    seg = 16384
    if widths == "reduced":
        cfg["model"].update({"inter_channels": 16, "upsample_initial_channel": 32,
                             "multi_period_discriminator_periods": [2, 3]})
        cfg["data"]["n_mel_channels"] = 40
        cfg["train"]["segment_size"] = seg = 4096
    state = copy.deepcopy(VocoderGAN(**cfg).state_dict())
    batch = synthetic.vocoder_batch(2, cfg["model"]["inter_channels"], segment_size=seg, seed=3)
    ops.set_deterministic(True)
    try:
        a, la = _grads(gpu, cfg, state, batch)
        b, lb = _grads(gpu, cfg, state, batch)
    finally:
        ops.set_deterministic(False)
    for idx, name in ((0, "generator"), (1, "discriminators")):
        assert a[idx].abs().sum() > 0
        same = torch.equal(a[idx], b[idx])
        if not same:
            d = (a[idx] - b[idx]).abs()
            raise AssertionError("%s gradients differ between two identical steps: %d of %d elements, max |diff| %.3e"
                                 % (name, int((d > 0).sum()), d.numel(), float(d.max())))
