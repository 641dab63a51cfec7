"""Shared helpers for golden fixtures: deterministic parameter/input fill from a NumPy
default_rng stream (stable across torch versions and machines, so the GPU box can regenerate
full-width weights without receiving any reference file)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rng_tensor(rng, shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(tuple(shape)) * scale).astype(np.float32))


def fill_state_dict(keys_shapes, seed):
    """keys_shapes: ordered list of (key, shape).  weight_g ~ 1 + 0.1 N, biases / LayerNorm
    beta ~ 0.1 N, gamma ~ 1 + 0.1 N, everything else ~ N(0, 1) * fan-in^-1/2 (weight_v: N(0,1),
    it is normalised anyway)."""
    import zlib
    sd = {}
    for key, shape in keys_shapes:
        # one independent stream per key: the fill does not depend on parameter registration order
        rng = np.random.default_rng([int(seed), zlib.crc32(key.encode())])
        shape = tuple(shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf in ("weight_g", "gamma"):
            t = 1.0 + 0.1 * rng_tensor(rng, shape)
        elif leaf in ("bias", "beta"):
            t = 0.1 * rng_tensor(rng, shape)
        elif leaf == "weight_v":
            t = rng_tensor(rng, shape)
        else:
            fan = 1
            for d in shape[1:]:
                fan *= d
            if len(shape) == 1:
                fan = 1
            t = rng_tensor(rng, shape, scale=max(fan, 1) ** -0.5)
        sd[key] = t
    return sd


def keys_shapes_of(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items()]


def checksum(t, n_samples=16, seed=0):
    """(sum, abs-sum, sampled elements) of a tensor, as float64 numpy."""
    t = t.detach().double().cpu().reshape(-1)
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, t.numel(), size=n_samples)
    return np.array([t.sum().item(), t.abs().sum().item()]), idx, t[torch.from_numpy(idx)].numpy()


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
