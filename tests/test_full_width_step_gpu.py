"""GPU: ONE whole reference batch at FULL widths (configs/base.json and configs/48k_base.json, B = 2) on the HIP path
against the CPU oracle trainer -- both losses and every parameter gradient of both optimizer passes.  The
reduced-width version of the same comparison is tests/test_training_step_gpu.py; this one exercises the tile
variants, split reductions and grouped kernels the real channel counts select (a base-width B = 2 oracle batch costs
about a second of host time on the GPU box).  Each batch is also run through the oracle in FLOAT64, and both fp32 steps are ranked
by their distance to it (golden_util.rank_against_f64)."""
import copy

import pytest
import torch

from golden_util import close_kinked, rank_against_f64

pytestmark = pytest.mark.gpu


def _compare(module, trainer, batch, gpu, tol_loss=2e-4, tol_grad=5e-4, trainer64=None, name=""):
    lc = trainer.batch(batch)
    names = {id(p): n for n, p in module.named_parameters()}
    grads = {}

    def probe(idx, opt):
        for p in opt.params:
            grads[names[id(p)]] = p.grad.detach().cpu().clone()

    out = module.fit_batch({k: v.to(gpu) for k, v in batch.items()}, after_backward=probe)
    for a, b, n in zip((out["g"], out["d"]), lc, ("loss_g", "loss_d")):
        assert abs(float(a) - float(b)) <= tol_loss * abs(float(b)) + 1e-5, (n, float(a), float(b))
    ref = dict(trainer.grads_g)
    ref.update(trainer.grads_d)
    assert set(ref) <= set(grads), set(ref) - set(grads)
    # per-tensor comparison with the kink-aware statistic (golden_util.close_kinked); tensors whose gradient is
    # analytically ~0 (e.g. softmax key biases) get an absolute floor from the largest gradient of their own network
    tops = {}
    for kk, v in ref.items():
        tops[kk.split(".")[0]] = max(tops.get(kk.split(".")[0], 0.0), float(v.abs().max()))
    for k, b in ref.items():
        close_kinked(k, grads[k], b, tol=tol_grad, floor=2e-6 * tops[k.split(".")[0]])
    if trainer64 is not None:
        l64 = trainer64.batch(batch)
        ref64 = dict(trainer64.grads_g)
        ref64.update(trainer64.grads_d)
        rows, nets = rank_against_f64(name, grads, ref, ref64)
        for a, b32, b64, n in zip((out["g"], out["d"]), lc, l64, ("loss_g", "loss_d")):
            # the losses against float64: HIP no further than 4 x torch-CPU fp32 (+ 2e-6 relative: one fp32 rounding of the sum)
            eh, ec = abs(float(a) - float(b64)), abs(float(b32) - float(b64))
            assert eh <= 4 * ec + 2e-6 * abs(float(b64)), (name, n, float(a), float(b32), float(b64))
        return rows, nets, (out, lc, l64)


@pytest.mark.parametrize("config", ["base", "48k"])
def test_vocoder_gan_full_width(gpu, config):
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import configs, synthetic
    from vcvits_amd.light.vcvits import DEFAULT_PERIODS, VocoderGAN
    torch.manual_seed(0)
    cfg = configs.base() if config == "base" else configs.base_48k()
    periods = cfg["model"].get("multi_period_discriminator_periods") or DEFAULT_PERIODS
    module = VocoderGAN(**cfg)
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, periods, vocoder_only=True)
    trainer64 = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, periods, vocoder_only=True, dtype=torch.float64)
    module = module.to(gpu)
    module.configure_optimizers()
    _compare(module, trainer, synthetic.vocoder_batch(2, cfg["model"]["inter_channels"], seed=21), gpu, trainer64=trainer64,
             name="vocoder/" + config)


@pytest.mark.parametrize("config", ["base", "48k"])
def test_vcvits_full_width(gpu, config):
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import configs, synthetic
    from vcvits_amd.light.vcvits import DEFAULT_PERIODS, VCVITS
    torch.manual_seed(1)
    cfg = configs.base() if config == "base" else configs.base_48k()
    cfg["model"]["p_dropout"] = 0.0  # the two sides draw dropout masks from different generators
    periods = cfg["model"].get("multi_period_discriminator_periods") or DEFAULT_PERIODS
    module = VCVITS(**cfg)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if ".post." in n:
                p.normal_(0.0, 0.02)  # zero-initialised coupling layers would be identities
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, periods, vocoder_only=False)
    trainer64 = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, periods, vocoder_only=False, dtype=torch.float64)
    module = module.to(gpu)
    module.configure_optimizers()
    m = cfg["model"]
    batch = synthetic.full_batch(2, m["hubert_channels"], t_y=96, t_x=52, seed=22)
    batch["y_wav_lengths"][1] = 80 * 512
    batch["y_wav_values"][1, :, 80 * 512:] = 0
    batch["x_hubert_features_lengths"][1] = 44
    batch["x_pitch_lengths"][1] = 44
    g = torch.Generator().manual_seed(23)
    batch["noise"] = torch.randn(2, m["inter_channels"], 96, generator=g)
    batch["ids_slice"] = torch.tensor([7, 41])
    _compare(module, trainer, batch, gpu, trainer64=trainer64, name="full/" + config)
