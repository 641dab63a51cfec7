"""Checkpoint container compatible with what the reference's Lightning run writes and reads
(SURVEY.md section 5 / 8f rank 2): a dict with `state_dict` (keys `net_g.*`, `net_period_d.*`,
`net_scale_d.*` with old-style `weight_g` / `weight_v` pairs), `optimizer_states`, `hyper_parameters`
(saved by `save_hyperparameters`, vcvits.py:31), `epoch`, `global_step`; the newest run's
`checkpoints/last.ckpt` is picked the way train.py:39-48 does, the lexicographically last `*.ckpt`
the way infer.py:13-14 does; loading goes through `VCVITS.on_load_checkpoint` (vcvits.py:265-282:
shape-mismatched tensors keep the fresh values, unknown keys are dropped, optimizer state is
discarded when anything changed)."""
import glob
import os
from typing import Optional

import torch


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


def save_checkpoint(module, path: str) -> str:
    """Write `path` (e.g. .../checkpoints/last.ckpt) atomically."""
    ckpt = {
        "state_dict": {k: v.detach().cpu() for k, v in module.state_dict().items()},
        "hyper_parameters": module.hparams.to_dict(),
        "epoch": int(module.current_epoch),
        "global_step": int(module.global_step),
        "optimizer_states": [],
        "format": "vcvits_amd.flat_adamw.v1",
    }
    for opt in (module.optim_g, module.optim_d):
        if opt is not None:
            sd = opt.state_dict()
            ckpt["optimizer_states"].append({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sd.items()})
    tmp = path + ".tmp"
    torch.save(ckpt, tmp)
    os.replace(tmp, path)
    return path


def load_checkpoint(module, path: str, map_location="cpu") -> dict:
    """Tolerant load (reference semantics): returns the checkpoint dict after `on_load_checkpoint`."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    module.on_load_checkpoint(ckpt)
    own = module.state_dict()
    filtered = {k: v for k, v in ckpt["state_dict"].items() if k in own}
    with torch.no_grad():
        for k, v in filtered.items():
            own[k].copy_(v.to(own[k].device))  # in place: parameters may be views of a flat optimizer buffer
    module.current_epoch = int(ckpt.get("epoch", 0))
    module.global_step = int(ckpt.get("global_step", 0))
    states = ckpt.get("optimizer_states")
    if states and module.optim_g is not None and len(states) == 2 and ckpt.get("format") == "vcvits_amd.flat_adamw.v1":
        for opt, sd in zip((module.optim_g, module.optim_d), states):
            if sd["exp_avg"].numel() == opt.numel:
                opt.load_state_dict({k: (v.to(opt.flat.device) if torch.is_tensor(v) else v) for k, v in sd.items()})
    return ckpt
