"""GPU: ONE whole reference batch at FULL widths (configs/base.json and configs/48k_base.json, B = 2) on the HIP path
against the CPU oracle trainer -- both losses and every parameter gradient of both optimizer passes.  The
reduced-width version of the same comparison is tests/test_training_step_gpu.py; this one exercises the tile
variants, split reductions and grouped kernels the real channel counts select (a base-width B = 2 oracle batch costs
about a second of host time on the GPU box).  Each batch is also run through the oracle in FLOAT64, and both fp32 steps are ranked
by their distance to it (_rank_against_f64)."""
import copy

import pytest
import torch

from golden_util import close_kinked

pytestmark = pytest.mark.gpu


def _rank_against_f64(name, grads, ref32, ref64):
    """Ranking against float64 (verdict r5 #6): is the HIP step any further from the truth than torch-CPU fp32 is?

    Per tensor and per network: relative L2 distance to the float64 oracle step of (a) the HIP step, (b) the torch-CPU fp32
    oracle step (tensors whose gradient is analytically ~0 get an absolute floor from their network's largest gradient).
    What the MI355X shows (profiles/r6_f64_ranking.txt, 2,722 tensors of four full-width batches): the two fp32 steps are
    statistically the SAME distance from float64 -- median hip / cpu32 ratio 0.44 .. 2.2 per batch, each side has whole
    sub-networks at ~2e-4 where the other sits at 3e-7 (ONE leaky-ReLU kink flipped on that side: a pre-activation within
    fp32 rounding of zero), worst tensor 4.4e-3 (HIP) / 3.5e-3 (CPU fp32), worst network 1.1e-4 / 1.1e-4.  A per-tensor
    "HIP <= 2 x CPU" cannot hold for ANY pair of fp32 implementations (CPU fp32 fails it against HIP on 43 .. 374 tensors per
    batch, HIP against CPU on 35 .. 256), so the assertions are the symmetric ones:
      * every tensor within 1e-2 of float64 (relative L2; observed worst 4.4e-3 HIP / 3.5e-3 CPU fp32) -- a wrong tile or a
        dropped term is 1e-1 and up;
      * every network's whole gradient within 5e-4;
      * HIP's count of tensors further than 1e-4 from float64 at most twice CPU fp32's count plus 5 % of the tensors;
      * median hip / cpu32 ratio <= 3.
    Rows are appended to $VCVITS_RANK_STATS when set."""
    import os
    import statistics
    tops = {}
    for k, v in ref64.items():
        tops[k.split(".")[0]] = max(tops.get(k.split(".")[0], 0.0), float(v.abs().max()))
    rows, nets = [], {}
    for k, r64 in ref64.items():
        r64 = r64.double()
        den = r64.norm().item() + 2e-6 * tops[k.split(".")[0]] * r64.numel() ** 0.5
        eh = (grads[k].double() - r64).norm().item()
        ec = (ref32[k].double() - r64).norm().item()
        rows.append((k, r64.numel(), eh / den, ec / den))
        n = nets.setdefault(k.split(".")[0], [0.0, 0.0, 0.0])
        n[0] += eh * eh
        n[1] += ec * ec
        n[2] += r64.norm().item() ** 2
    nets = {net: ((a / c) ** 0.5, (b / c) ** 0.5) for net, (a, b, c) in nets.items()}
    path = os.environ.get("VCVITS_RANK_STATS")
    if path:
        with open(path, "a") as f:
            for k, n, eh, ec in rows:
                f.write("%s %s n=%d hip=%.3e cpu32=%.3e ratio=%.2f\n" % (name, k, n, eh, ec, eh / (ec + 1e-300)))
            for net, (a, b) in nets.items():
                f.write("%s NET %s hip=%.3e cpu32=%.3e ratio=%.2f\n" % (name, net, a, b, a / (b + 1e-300)))
    for k, n, eh, ec in rows:
        assert eh <= 1e-2, "%s %s: HIP gradient %.3e from float64 (CPU fp32: %.3e)" % (name, k, eh, ec)
    for net, (a, b) in nets.items():
        assert a <= 5e-4, "%s %s: whole gradient %.3e from float64 (CPU fp32: %.3e)" % (name, net, a, b)
    far_h = sum(1 for _, _, eh, _ in rows if eh > 1e-4)
    far_c = sum(1 for _, _, _, ec in rows if ec > 1e-4)
    assert far_h <= 2 * far_c + 0.05 * len(rows), (name, far_h, far_c, len(rows))
    med = statistics.median(eh / (ec + 1e-30) for _, _, eh, ec in rows)
    assert med <= 3.0, (name, med)
    return rows, nets


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
        rows, nets = _rank_against_f64(name, grads, ref, ref64)
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
