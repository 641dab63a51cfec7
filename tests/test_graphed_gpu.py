"""GPU: the discriminator step's no-grad generator pass replayed from a HIP graph (vcvits_amd/light/graphed.py; the
reference runs that pass eagerly under torch.no_grad(): vits/light/vcvits.py:119,153).  The replayed pass must be the
eager pass: same losses step for step from identical state; fresh dropout masks on every replay."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(p_dropout):
    from vcvits_amd import configs
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 32, "hidden_channels": 32, "filter_channels": 64, "n_heads": 2,
                         "upsample_initial_channel": 64, "hubert_channels": 48, "gin_channels": 16, "p_dropout": p_dropout,
                         "multi_period_discriminator_periods": [2, 3]})
    cfg["data"]["n_mel_channels"] = 40
    return cfg


def _batches(cfg, n, gpu, with_draws):
    from vcvits_amd import synthetic
    m = cfg["model"]
    out = []
    for i in range(n):
        b = synthetic.full_batch(4, m["hubert_channels"], seed=50 + i)
        if with_draws:
            g = torch.Generator().manual_seed(90 + i)
            b["noise"] = torch.randn(4, m["inter_channels"], 384, generator=g)
            b["ids_slice"] = torch.tensor([5, 100, 17, 200])
        out.append({k: v.to(gpu) for k, v in b.items()})
    return out


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_graphed_generator_pass_equals_eager(gpu, dtype):
    from vcvits_amd import ops
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VCVITS
    cfg = _cfg(0.0)
    torch.manual_seed(11)
    ref = VCVITS(**cfg)
    sd = copy.deepcopy(ref.state_dict())
    batches = _batches(cfg, 2, gpu, with_draws=True)
    losses = {}
    ops.set_compute_dtype(dtype)
    try:
        for mode in (False, True):
            graphed.set_enabled(mode)
            torch.manual_seed(12)
            mod = VCVITS(**cfg)
            mod.load_state_dict(sd)
            mod = mod.to(gpu)
            mod.configure_optimizers()
            ls = []
            for i in range(8):
                out = mod.fit_batch(batches[i % 2])
                ls.append((float(out["g"]), float(out["d"])))
            losses[mode] = ls
            if mode:
                g = mod.__dict__["_g_graph"]
                assert not g.failed and g.replays >= 4, (g.failed, g.replays)
            mod.optim_g.close()
            mod.optim_d.close()
    finally:
        graphed.set_enabled(True)
        ops.set_compute_dtype("f32")
    tol = 2e-5 if dtype == "f32" else 2e-3  # (weight-gradient atomics / split orders move the parameters in the last bits)
    for (g0, d0), (g1, d1) in zip(losses[False], losses[True]):
        assert abs(g0 - g1) <= tol * abs(g0) and abs(d0 - d1) <= tol * abs(d0), (losses[False], losses[True])


def test_graph_replays_draw_fresh_dropout_masks(gpu):
    """A captured sequence bakes the host seed into its kernel arguments; the device-side seed offset bumped before every
    replay must give each replay its own masks (forward dropout and the attention kernel's dropout)."""
    from vcvits_amd import ops
    from vcvits_amd.light import graphed

    def fn(b):
        y = ops.dropout(b["x"], 0.5, True)
        o, _ = ops.rel_attention(b["q"], b["q"], b["q"], b["ek"], b["ek"], b["mask"], 2, 4, pdrop=0.5, training=True, want_attn=False)
        return y, o

    g = graphed.GraphedNoGrad(fn, warmup=2)
    gen = torch.Generator().manual_seed(3)
    batch = {"x": torch.randn(4, 32, 200, generator=gen), "q": torch.randn(2, 64, 120, generator=gen),
             "ek": torch.randn(9, 32, generator=gen) * 0.1, "mask": torch.ones(2, 120)}
    batch = {k: v.to(gpu) for k, v in batch.items()}
    graphed.set_enabled(True)
    outs = []
    with torch.no_grad():
        for _ in range(7):
            y, o = g(batch)
            outs.append((y.clone(), o.clone()))
    assert not g.failed and g.replays >= 4, (g.failed, g.replays)
    for y, o in outs:
        assert bool(torch.isfinite(o).all())
        kept = (y != 0).float().mean().item()
        assert 0.4 < kept < 0.6, kept
    for i in range(len(outs)):
        for j in range(i + 1, len(outs)):
            assert not torch.equal(outs[i][0], outs[j][0]) and not torch.equal(outs[i][1], outs[j][1]), (i, j)
