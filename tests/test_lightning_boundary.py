"""Boundary with the reference's trainer (CPU; no GPU, no Lightning install needed).

The reference's training module is a `pytorch_lightning.LightningModule` (vits/light/vcvits.py:14,28) that train.py hands to
`Trainer.fit` (train.py:85,110-113), which type-checks it and the optimizers `configure_optimizers` returns.  Lightning is not
in this image, so the check runs against a STUB `pytorch_lightning` module that reproduces the parts of LightningModule that
constrain a subclass: read-only `hparams` / `current_epoch` / `global_step` properties and `save_hyperparameters` collecting
the constructor's kwargs from the caller's frame."""
import importlib
import inspect
import sys
import types

import torch
from torch import nn


def _stub_lightning():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self._trainer = None
            self._hparams = None

        def save_hyperparameters(self, *names):
            frame = inspect.currentframe().f_back
            init_args = dict(frame.f_locals.get("kwargs", {}))
            self._hparams = types.SimpleNamespace(**{n: init_args[n] for n in names})

        @property
        def hparams(self):  # (no setter: assigning self.hparams raises, as in Lightning)
            return self._hparams

        @property
        def current_epoch(self):
            return self._trainer.current_epoch if self._trainer is not None else 0

        @property
        def global_step(self):
            return self._trainer.global_step if self._trainer is not None else 0

        @property
        def logger(self):
            return self._trainer.logger if self._trainer is not None else None

    pl.LightningModule = LightningModule
    return pl


def _small_cfg():
    from vcvits_amd import configs
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 8, "hidden_channels": 8, "filter_channels": 16, "n_heads": 2, "n_layers": 1,
                         "upsample_initial_channel": 16, "hubert_channels": 12, "gin_channels": 8,
                         "multi_period_discriminator_periods": [2, 3]})
    cfg["data"]["n_mel_channels"] = 20
    return cfg


def test_vcvits_is_a_lightning_module_where_lightning_imports():
    saved = {k: sys.modules.get(k) for k in ("pytorch_lightning", "vcvits_amd.light.vcvits")}
    pl = _stub_lightning()
    sys.modules["pytorch_lightning"] = pl
    sys.modules.pop("vcvits_amd.light.vcvits", None)
    try:
        mod = importlib.import_module("vcvits_amd.light.vcvits")
        assert mod.HAS_LIGHTNING and issubclass(mod.VCVITS, pl.LightningModule) and issubclass(mod.VocoderGAN, pl.LightningModule)
        from vcvits_amd.hparams import HParams
        cfg = _small_cfg()
        m = mod.VCVITS(**HParams(**cfg))  # train.py:85: VCVITS(**hparams)
        assert isinstance(m, pl.LightningModule)
        # save_hyperparameters took the constructor's kwargs (vcvits.py:31); nested sections read by attribute
        assert m.hparams.data.hop_length == cfg["data"]["hop_length"] and m.hparams.train.segment_size == cfg["train"]["segment_size"]
        assert [n for n, _ in m.named_children()][:3] == ["net_g", "net_period_d", "net_scale_d"]
        # the hook Lightning-1.x's automatic optimisation calls once per optimizer (vcvits.py:54)
        assert list(inspect.signature(m.training_step).parameters) == ["batch", "batch_idx", "optimizer_idx"]
        opts, scheds = m.configure_optimizers()  # vcvits.py:247-263: ([optim_g, optim_d], [scheduler_g, scheduler_d])
        assert len(opts) == 2 and len(scheds) == 2
        for o in opts:
            assert isinstance(o, torch.optim.Optimizer) and o.param_groups[0]["lr"] == cfg["train"]["learning_rate"]
        for sch in scheds:
            assert all(hasattr(sch, a) for a in ("step", "state_dict", "load_state_dict", "get_last_lr"))
        # own counters while no Trainer is attached; the Trainer's once one is
        assert m.current_epoch == 0 and m.global_step == 0
        m.on_epoch_end()
        assert m.current_epoch == 1
        m._trainer = types.SimpleNamespace(current_epoch=7, global_step=1234, logger=None)
        assert m.current_epoch == 7 and m.global_step == 1234
        # an external scheduler writing the group's rate is what the next step uses (Lightning steps torch schedulers)
        opts[0].param_groups[0]["lr"] = 1e-5
        for o in opts:
            o.close()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        if saved["vcvits_amd.light.vcvits"] is not None:
            import vcvits_amd.light as light_pkg
            light_pkg.vcvits = saved["vcvits_amd.light.vcvits"]


def test_vcvits_is_a_plain_module_without_lightning():
    from vcvits_amd.light import vcvits as mod
    if mod.HAS_LIGHTNING:  # (an image that has Lightning: the other test's subject)
        return
    assert issubclass(mod.VCVITS, nn.Module)
    m = mod.VCVITS(**_small_cfg())
    assert m.hparams.data.hop_length == 512 and m.current_epoch == 0 and m.global_step == 0
    m.current_epoch = 3
    assert m.current_epoch == 3


def test_flat_adamw_step_runs_a_closure_first():
    """Lightning's automatic optimisation passes zero_grad + training_step + backward to `optimizer.step(closure=...)`."""
    from vcvits_amd.light.optim import FlatAdamW
    net = nn.Linear(3, 2)
    opt = FlatAdamW(net.parameters(), 1e-2)
    calls = []

    def closure():
        calls.append(1)
        opt.zero_grad()
        loss = net(torch.ones(1, 3)).sum()
        loss.backward()
        return loss

    try:
        opt.step(closure)
    except RuntimeError as e:  # the update kernel itself has no CPU fallback: everything before it ran
        assert "not on the GPU" in str(e)
    assert calls == [1] and opt.step_count == 1 and bytes(opt._touched) == b"\x01\x01"
    opt.close()
