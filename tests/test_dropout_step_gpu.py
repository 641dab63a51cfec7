"""GPU: the full-model training step at REAL widths with the configs' dropout (p_dropout = 0.1: attention probabilities,
attention / FFN outputs, FFN hidden -- relative_attention_transformer.py:40,44,173,304) against the oracle.  The two sides
draw masks from different generators, so the HIP step's draws are traced (ops.DROPOUT_TRACE: the masks are functions of
(seed, flat index)), regenerated, and handed to the oracle, which applies them in the reference's call order: same
elements dropped on both sides, losses and every gradient compared as in tests/test_full_width_step_gpu.py."""
import copy

import pytest
import torch

from golden_util import close_kinked

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config", ["base", "48k"])
def test_vcvits_full_width_with_dropout(gpu, config):
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import configs, ops, synthetic
    from vcvits_amd.light.vcvits import DEFAULT_PERIODS, VCVITS
    torch.manual_seed(1)
    cfg = configs.base() if config == "base" else configs.base_48k()
    assert cfg["model"]["p_dropout"] == 0.1  # the configs' own value
    periods = cfg["model"].get("multi_period_discriminator_periods") or DEFAULT_PERIODS
    module = VCVITS(**cfg)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if ".post." in n:
                p.normal_(0.0, 0.02)
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, periods, vocoder_only=False)
    module = module.to(gpu)
    module.configure_optimizers()
    m = cfg["model"]
    batch = synthetic.full_batch(2, m["hubert_channels"], t_y=96, t_x=52, seed=42)
    batch["x_hubert_features_lengths"][1] = 44
    batch["x_pitch_lengths"][1] = 44
    g = torch.Generator().manual_seed(43)
    batch["noise"] = torch.randn(2, m["inter_channels"], 96, generator=g)
    batch["ids_slice"] = torch.tensor([7, 41])

    names = {id(p): n for n, p in module.named_parameters()}
    grads = {}

    def probe(idx, opt):
        for p in opt.params:
            grads[names[id(p)]] = p.grad.detach().cpu().clone()

    ops.DROPOUT_TRACE[0] = trace = []
    try:
        out = module.fit_batch({k: v.to(gpu) for k, v in batch.items()}, after_backward=probe)
    finally:
        ops.DROPOUT_TRACE[0] = None
    torch.cuda.synchronize()
    # 3 layers x (attention probabilities, attention output, FFN hidden, FFN output), generator pass + discriminator pass
    assert len(trace) == 2 * m["n_layers"] * 4 and [t[0] for t in trace[:4]] == ["attn", "drop", "drop", "drop"], trace[:5]
    masks = [ops.dropout_mask(shape, p, seed, gpu).cpu() for (_, p, seed, shape) in trace]
    kept = float(torch.cat([mk.reshape(-1) for mk in masks]).ne(0).float().mean())
    assert 0.88 < kept < 0.92, kept
    it = iter(masks)

    def drop(t):
        mk = next(it)
        return t * mk.reshape(t.shape)

    trainer.drop = drop
    lc = trainer.batch(batch)
    assert next(it, None) is None, "the oracle consumed fewer dropout draws than the HIP step made"
    for a, b, n in zip((out["g"], out["d"]), lc, ("loss_g", "loss_d")):
        assert abs(float(a) - float(b)) <= 2e-4 * abs(float(b)) + 1e-5, (n, float(a), float(b))
    ref = dict(trainer.grads_g)
    ref.update(trainer.grads_d)
    tops = {}
    for kk, v in ref.items():
        tops[kk.split(".")[0]] = max(tops.get(kk.split(".")[0], 0.0), float(v.abs().max()))
    for k, b in ref.items():
        close_kinked(k, grads[k], b, tol=5e-4, floor=2e-6 * tops[k.split(".")[0]])
