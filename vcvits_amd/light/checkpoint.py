"""Checkpoints in the layout the reference's Lightning run writes and reads.

A Lightning `.ckpt` of the reference module (`Trainer.save_checkpoint` on `VCVITS`, vcvits.py:28-283) is a
pickled dict:
  `state_dict`         module tree keys: `net_g.*`, `net_period_d.*`, `net_scale_d.*` (old-style `weight_g` /
                       `weight_v` pairs), plus third-party entries this build has no tensor for:
                       `net_g.enc_p.hubert.*` (the frozen fairseq model) and `audio_pipeline.*` (torchaudio windows);
  `optimizer_states`   one torch.optim state_dict per optimizer, in `configure_optimizers` order (vcvits.py:247-257):
                       {"state": {param_index: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{"lr", "betas",
                       "eps", "weight_decay", "params": [indices]}]}; indices count `net_g.parameters()` (or the
                       chained discriminator parameters) in registration order, frozen parameters included;
  `lr_schedulers`      one ExponentialLR state_dict per scheduler (`last_epoch`, `_last_lr`, `gamma`, ...);
  `hyper_parameters`   what `save_hyperparameters` captured (vcvits.py:31), `epoch`, `global_step`,
                       `pytorch-lightning_version`.
`save_checkpoint` writes that layout (AdamW moments scattered back from the flat buffers into per-parameter
entries), `load_checkpoint` reads it -- and this build's round-1 container (`format: vcvits_amd.flat_adamw.v1`) --
through `VCVITS.on_load_checkpoint` (vcvits.py:265-282: shape-mismatched tensors keep the fresh values, unknown keys
are dropped, optimizer state is discarded when anything changed).  Discovery: the newest run's
`checkpoints/last.ckpt` as train.py:39-48 picks it, the lexicographically last `*.ckpt` as infer.py:13-14 does."""
import glob
import logging
import os
from typing import Optional

import torch

# state_dict entries of the reference module that belong to third-party sub-modules this build does not hold
# tensors for; they are skipped on load without counting as a change of the model
THIRD_PARTY_PREFIXES = ("net_g.enc_p.hubert.", "audio_pipeline.")
_BUFFER_SUFFIXES = (".window", ".num_batches_tracked", ".running_mean", ".running_var")
FORMAT_V1 = "vcvits_amd.flat_adamw.v1"


def last_checkpoint(path: str) -> Optional[str]:
    """train.py:39-48: <path>/lightning_logs/version_<N>/checkpoints/last.ckpt of the highest N."""
    root = os.path.join(path, "lightning_logs")
    if not os.path.exists(root):
        return None
    versions = glob.glob(os.path.join(root, "version_*"))
    if not versions:
        return None
    last_ver = sorted(versions, key=lambda p: int(p.split("_")[-1]))[-1]
    ckpt = os.path.join(last_ver, "checkpoints", "last.ckpt")
    return ckpt if os.path.exists(ckpt) else None


def newest_ckpt_in(directory: str) -> Optional[str]:
    """infer.py:13-14: the lexicographically last *.ckpt of a directory."""
    files = sorted(glob.glob(os.path.join(directory, "*.ckpt")))
    return files[-1] if files else None


def next_version_dir(path: str) -> str:
    root = os.path.join(path, "lightning_logs")
    os.makedirs(root, exist_ok=True)
    nums = [int(p.split("_")[-1]) for p in glob.glob(os.path.join(root, "version_*"))]
    d = os.path.join(root, "version_%d" % (max(nums) + 1 if nums else 0), "checkpoints")
    os.makedirs(d, exist_ok=True)
    return d


# ---------------------------------------------------------------------------------------------------------
def _optimizer_prefixes(idx):
    return ("net_g.",) if idx == 0 else ("net_period_d.", "net_scale_d.")


def _module_param_names(module, idx):
    """Parameter names in the order torch hands them to the reference's optimizer idx (vcvits.py:248-257)."""
    names = []
    for pre in _optimizer_prefixes(idx):
        sub = getattr(module, pre[:-1])
        names += [pre + n for n, _ in sub.named_parameters()]
    return names


def _torch_optim_state(module, opt, idx):
    """FlatAdamW -> torch.optim.AdamW.state_dict() layout (per-parameter moments copied out of the flat buffers)."""
    names = _module_param_names(module, idx)
    by_id = {id(p): (i, o) for i, (p, o) in enumerate(zip(opt.params, opt.offsets))}
    params = dict(module.named_parameters())
    state = {}
    for j, name in enumerate(names):
        p = params[name]
        i, o = by_id[id(p)]
        if opt._pstep[i] == 0:
            continue  # torch creates a parameter's state at its first step
        n = p.numel()
        state[j] = {"step": torch.tensor(float(opt._pstep[i])),
                    "exp_avg": opt.exp_avg[o:o + n].view(p.shape).detach().cpu().clone(),
                    "exp_avg_sq": opt.exp_avg_sq[o:o + n].view(p.shape).detach().cpu().clone()}
    group = {"lr": opt.lr, "betas": tuple(opt.betas), "eps": opt.eps, "weight_decay": opt.weight_decay,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "initial_lr": opt.base_lr, "params": list(range(len(names)))}
    return {"state": state, "param_groups": [group]}


def save_checkpoint(module, path: str) -> str:
    """Write `path` (e.g. .../checkpoints/last.ckpt) atomically, in the Lightning layout described above."""
    from .. import ops
    ops.check_indices()  # (a run that looked up embedding rows outside their table trained on zeros: do not save it silently)
    ckpt = {
        "epoch": int(module.current_epoch),
        "global_step": int(module.global_step),
        "pytorch-lightning_version": "2.0.2",  # requirements.txt of the reference; readers only look at the layout
        "state_dict": {k: v.detach().cpu() for k, v in module.state_dict().items()},
        "optimizer_states": [],
        "lr_schedulers": [],
        "hparams_name": "kwargs",
        "hyper_parameters": module.hparams.to_dict(),
        "vcvits_amd": {"dropout_seed_state": ops.get_seed_state()},
    }
    for idx, (opt, sch) in enumerate(((module.optim_g, getattr(module, "scheduler_g", None)),
                                      (module.optim_d, getattr(module, "scheduler_d", None)))):
        if opt is not None:
            ckpt["optimizer_states"].append(_torch_optim_state(module, opt, idx))
            if sch is not None:
                sd = sch.state_dict()
                sd.update({"base_lrs": [opt.base_lr], "_step_count": sd["last_epoch"] + 1, "verbose": False})
                ckpt["lr_schedulers"].append(sd)
    tmp = path + ".tmp"
    torch.save(ckpt, tmp)
    os.replace(tmp, path)
    return path


def _read(path, map_location, allow_pickle):
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:  # objects outside torch's allow-list (e.g. a pickled HParams / AttributeDict)
        if not allow_pickle:
            raise RuntimeError("%s holds pickled objects torch.load(weights_only=True) refuses (%s); pass "
                               "allow_pickle=True only for checkpoints you trust" % (path, type(e).__name__)) from e
        return torch.load(path, map_location=map_location, weights_only=False)


def _plan_torch_optim(module, opt, idx, ckpt, st):
    """Align a torch.optim.AdamW state (keyed by parameter index in the CHECKPOINT's parameter order) with the flat
    buffers of `opt`, WITHOUT touching them.  Returns the list of (parameter slot, offset, numel, state entry) to copy,
    or None when the checkpoint's parameter list cannot be aligned by name or a moment has the wrong shape."""
    pres = _optimizer_prefixes(idx)
    ck_names = [k for k in ckpt["state_dict"] if k.startswith(pres) and not k.endswith(_BUFFER_SUFFIXES)]
    pids = [pid for g in st["param_groups"] for pid in g["params"]]
    if len(ck_names) != len(pids):
        logging.info("optimizer %d: %d parameter indices but %d candidate tensors in state_dict; state dropped",
                     idx, len(pids), len(ck_names))
        return None
    name_of = dict(zip(pids, ck_names))
    params = dict(module.named_parameters())
    where = {id(p): (i, o) for i, (p, o) in enumerate(zip(opt.params, opt.offsets))}
    plan = []
    for pid, s in st["state"].items():
        p = params.get(name_of.get(pid))
        if p is None or id(p) not in where:
            continue  # third-party (frozen HuBERT) entries
        if tuple(s["exp_avg"].shape) != tuple(p.shape) or tuple(s["exp_avg_sq"].shape) != tuple(p.shape):
            logging.info("optimizer %d: moment shape mismatch for %s; state dropped", idx, name_of.get(pid))
            return None
        i, o = where[id(p)]
        plan.append((i, o, p.numel(), s))
    return plan


def _commit_torch_optim(opt, st, plan):
    """Write a validated plan of _plan_torch_optim into the flat buffers."""
    opt.exp_avg.zero_()
    opt.exp_avg_sq.zero_()
    opt._pstep = [0] * len(opt.params)
    top = 0
    for i, o, n, s in plan:
        opt.exp_avg[o:o + n].copy_(s["exp_avg"].reshape(-1).to(opt.exp_avg.device))
        opt.exp_avg_sq[o:o + n].copy_(s["exp_avg_sq"].reshape(-1).to(opt.exp_avg.device))
        opt._pstep[i] = int(float(s["step"]))
        top = max(top, opt._pstep[i])
    opt.step_count = top
    g0 = st["param_groups"][0]
    opt.set_lr(g0["lr"])
    if "initial_lr" in g0:
        opt.base_lr = float(g0["initial_lr"])


def load_checkpoint(module, path: str, map_location="cpu", allow_pickle=False) -> dict:
    """Tolerant load (reference semantics): returns the checkpoint dict after `on_load_checkpoint`.  Restores
    parameters (in place: they may be views of a flat optimizer buffer), epoch / global_step, and -- when the
    optimizers exist (`configure_optimizers` was called) and `on_load_checkpoint` kept the optimizer states -- the
    AdamW moments, per-parameter step counts, learning rates and scheduler positions."""
    from .. import ops
    ckpt = _read(path, map_location, allow_pickle)
    order = list(ckpt["state_dict"])  # optimizer indices follow this order
    third = [k for k in order if k.startswith(THIRD_PARTY_PREFIXES)]
    held = {k: ckpt["state_dict"].pop(k) for k in third}  # not this build's tensors: not a "change" of the model
    module.on_load_checkpoint(ckpt)
    own = module.state_dict()
    with torch.no_grad():
        for k, v in ckpt["state_dict"].items():
            if k in own:
                own[k].copy_(v.to(own[k].device))
    for opt in (module.optim_g, module.optim_d):
        if opt is not None and opt.flat.is_cuda:
            ops.invalidate_weights(opt.flat.data_ptr(), opt.flat.data_ptr() + 4 * opt.numel)
    ckpt["state_dict"].update(held)
    ckpt["state_dict"] = {k: ckpt["state_dict"][k] for k in order if k in ckpt["state_dict"]}
    module.current_epoch = int(ckpt.get("epoch", 0))
    module.global_step = int(ckpt.get("global_step", 0))
    extra = ckpt.get("vcvits_amd") or {}
    if "dropout_seed_state" in extra:
        # the saved stream position is rank 0's; every other data-parallel rank keeps its own offset (as
        # configure_optimizers seeds it), or all ranks would draw identical dropout masks after a resume
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        ops.set_seed_state((int(extra["dropout_seed_state"]) + 0x9E3779B97F4A7C15 * rank) % (1 << 64))
    states = ckpt.get("optimizer_states")
    restored = False
    if states and module.optim_g is not None and len(states) == 2:
        if ckpt.get("format") == FORMAT_V1:
            opts = (module.optim_g, module.optim_d)
            if all(sd["exp_avg"].numel() == opt.numel for opt, sd in zip(opts, states)):
                for opt, sd in zip(opts, states):
                    opt.load_state_dict({k: (v.to(opt.flat.device) if torch.is_tensor(v) else v) for k, v in sd.items()})
                restored = True
        elif all(isinstance(s, dict) and "param_groups" in s for s in states):
            # both optimizers or neither: validate everything first, then write (a half-restored pair -- G with its
            # moments and rate, D fresh -- would train with inconsistent state)
            opts = (module.optim_g, module.optim_d)
            plans = [_plan_torch_optim(module, opt, i, ckpt, s) for i, (opt, s) in enumerate(zip(opts, states))]
            if all(pl is not None for pl in plans):
                for opt, s, pl in zip(opts, states, plans):
                    _commit_torch_optim(opt, s, pl)
                restored = True
    # schedulers: vcvits.py:258-261 re-seats last_epoch to current_epoch - 1 in configure_optimizers (which Lightning
    # calls before it restores the loop state, i.e. with current_epoch still 0); a restored scheduler state then
    # overrides it.  Without restored optimizer state the rate stays what configure_optimizers set -- the reference's
    # behaviour after on_load_checkpoint dropped `optimizer_states`.
    schs = [s for s in (getattr(module, "scheduler_g", None), getattr(module, "scheduler_d", None)) if s is not None]
    lrs = ckpt.get("lr_schedulers") or []
    for i, sch in enumerate(schs):
        sch.last_epoch = module.current_epoch - 1
        if restored and i < len(lrs):
            sch.load_state_dict(lrs[i])
    return ckpt
