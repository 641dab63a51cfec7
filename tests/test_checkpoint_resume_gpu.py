"""GPU: resume from a reference-shaped (Lightning) checkpoint and take ONE step -- weights, AdamW moments, per-parameter
step counts and learning rate restored by vcvits_amd.light.checkpoint (vits/light/vcvits.py:265-282 + Lightning's
optimizer restore) -- against the CPU oracle trainer resumed from the same state: both losses and the parameters AFTER
the optimizer step (which only agree if moments, step counts and rate were restored into the flat buffers correctly)."""
import copy

import pytest
import torch

from test_checkpoint_cpu import _full_module, _lightning_ckpt, _manifest

pytestmark = pytest.mark.gpu


def test_resume_then_step_matches_oracle(gpu, tmp_path):
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import synthetic
    from vcvits_amd.light import checkpoint as ck
    man = _manifest()
    w = man["widths"]
    raw = _lightning_ckpt(man, seed=4, lr=1.3e-4, epoch=2)
    for k in raw["state_dict"]:
        if k.endswith("weight_g"):
            raw["state_dict"][k] = raw["state_dict"][k].abs() + 0.5  # a trained-looking gain, not noise around zero
    p = str(tmp_path / "last.ckpt")
    torch.save(raw, p)
    torch.manual_seed(3)
    m = _full_module(man)
    m.hparams.model["p_dropout"] = 0.0
    for mod in m.modules():
        if hasattr(mod, "p_dropout"):
            mod.p_dropout = 0.0
    m = m.to(gpu)
    m.configure_optimizers()
    out = ck.load_checkpoint(m, p)
    assert "optimizer_states" in out and m.optim_g.lr == 1.3e-4 and m.optim_g.step_count == 41

    # the oracle, resumed from the same checkpoint: torch.optim.AdamW state by parameter, same rate
    cfg = m.hparams.to_dict()
    trainer = CpuTrainer({k: v.detach().cpu() for k, v in m.state_dict().items()}, cfg, w["PERIODS"], vocoder_only=False)
    params = dict(m.named_parameters())
    for opt_t, opt_m, plist in ((trainer.opt_g, m.optim_g, trainer.g_params), (trainer.opt_d, m.optim_d, trainer.d_params)):
        where = {id(q): o for q, o in zip(opt_m.params, opt_m.offsets)}
        names = {id(v): k for k, v in trainer.sd.items()}
        for q in plist:
            name = names[id(q)]
            o, n = where[id(params[name])], q.numel()
            opt_t.state[q] = {"step": torch.tensor(41.0),
                              "exp_avg": opt_m.exp_avg[o:o + n].view(q.shape).detach().cpu().clone(),
                              "exp_avg_sq": opt_m.exp_avg_sq[o:o + n].view(q.shape).detach().cpu().clone()}
        for g in opt_t.param_groups:
            g["lr"] = opt_m.lr

    batch = synthetic.full_batch(2, w["HUB"], t_y=96, t_x=52, seed=31)
    batch["x_pitch_values"] = batch["x_pitch_values"] % w["NPITCH"]
    batch["sid"] = batch["sid"] % w["NSPK"]
    gen = torch.Generator().manual_seed(32)
    batch["noise"] = torch.randn(2, w["C"], 96, generator=gen)
    batch["ids_slice"] = torch.tensor([5, 40])
    lg, ld = trainer.batch(batch)
    res = m.fit_batch({k: v.to(gpu) for k, v in batch.items()})
    torch.cuda.synchronize()
    assert abs(float(res["g"]) - float(lg)) <= 2e-4 * abs(float(lg)) + 1e-5
    assert abs(float(res["d"]) - float(ld)) <= 2e-4 * abs(float(ld)) + 1e-5
    # parameters after the step: AdamW's update is lr * m_hat / (sqrt(v_hat) + eps) -- with restored moments of size 1e-3 /
    # 1e-6 an update of ~1e-4 per element; compare the UPDATES (new - loaded), so a missed restore cannot hide behind the
    # unchanged bulk of the weights
    worst = 0.0
    for name, q in trainer.sd.items():
        if name not in params or not q.requires_grad and name not in params:
            continue
        loaded = raw["state_dict"][name].float()
        du_ref = q.detach() - loaded
        du_hip = params[name].detach().cpu() - loaded
        scale = float(du_ref.abs().max())
        if scale == 0.0:
            assert float(du_hip.abs().max()) == 0.0, name
            continue
        err = float((du_hip - du_ref).abs().max()) / scale
        worst = max(worst, err)
        assert err < 2e-2, (name, err, scale)  # the sign-like ratio m/sqrt(v) amplifies gradient noise where v is tiny
    assert worst > 0.0
