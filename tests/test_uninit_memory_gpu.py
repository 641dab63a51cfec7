"""GPU: no result of a training batch depends on the CONTENTS of uninitialised memory.

Why it is a test: a HIP-graph replay (vcvits_amd/light/graphed.py) sees, at every address it allocates, what the previous
replay left there -- where an eager pass sees whatever an earlier, unrelated tensor left.  A kernel that reads an element
nobody wrote (a masked-out frame, a tile edge) and multiplies it by a zero mask is invisible in eager runs and turns into
NaN the day the leftover is an inf.  Here every `torch.empty*` result on the GPU is pre-filled with NaN and one eager batch
of the full model and of the vocoder module must still give finite losses and parameter gradients equal to the un-poisoned
run's (tools/probes/poison_probe.py is the same at benchmark widths)."""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes"))


def _run(module, batch):
    grads = {}
    names = {id(p): n for n, p in module.named_parameters()}

    def probe(idx, opt):
        for p in opt.params:
            grads[names[id(p)]] = p.grad.detach().clone()
    out = module.fit_batch(batch, after_backward=probe)
    torch.cuda.synchronize()
    return {k: float(v) for k, v in out.items()}, grads


@pytest.mark.parametrize("workload", ["full", "vocoder"])
def test_results_do_not_depend_on_uninitialised_memory(gpu, workload):
    from poison_probe import Poison
    from vcvits_amd import configs, synthetic
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VCVITS, VocoderGAN
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 32, "hidden_channels": 32, "filter_channels": 64, "n_heads": 2, "p_dropout": 0.0,
                         "upsample_initial_channel": 64, "hubert_channels": 48, "gin_channels": 16,
                         "multi_period_discriminator_periods": [2, 3, 7]})
    cfg["data"]["n_mel_channels"] = 40
    m = cfg["model"]
    torch.manual_seed(21)
    cls = VCVITS if workload == "full" else VocoderGAN
    sd = copy.deepcopy(cls(**cfg).state_dict())
    if workload == "full":
        batch = synthetic.full_batch(5, m["hubert_channels"], seed=8, device=gpu)  # (odd batch: ragged tile edges)
        g = torch.Generator().manual_seed(9)
        batch["noise"] = torch.randn(5, m["inter_channels"], 384, generator=g).to(gpu)
        batch["ids_slice"] = torch.tensor([3, 250, 17, 100, 60], device=gpu)
    else:
        batch = synthetic.vocoder_batch(3, m["inter_channels"], seed=8, device=gpu)
    res = []
    graphed.set_enabled(False)
    try:
        for poisoned in (False, True):
            mod = cls(**cfg)
            mod.load_state_dict(sd)
            mod = mod.to(gpu).train()
            mod.configure_optimizers()
            if poisoned:
                with Poison():
                    res.append(_run(mod, batch))
            else:
                res.append(_run(mod, batch))
            mod.optim_g.close()
            mod.optim_d.close()
    finally:
        graphed.set_enabled(True)
    (l0, g0), (l1, g1) = res
    for k in l0:
        assert l1[k] == l1[k] and abs(l0[k] - l1[k]) <= 2e-5 * abs(l0[k]), (k, l0[k], l1[k])
    assert set(g0) == set(g1) and len(g0) > 100
    for n in g0:
        assert bool(torch.isfinite(g1[n]).all()), "gradient of %s holds NaN / inf with poisoned allocations" % n
        scale = float(g0[n].abs().max())
        assert float((g0[n] - g1[n]).abs().max()) <= 1e-3 * scale + 1e-6, n
