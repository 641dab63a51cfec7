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
    # spectral-normed layers (weight_orig + the power-iteration vectors weight_u / weight_v): unit vectors, as
    # torch.nn.utils.spectral_norm keeps them
    for key in list(sd):
        if key.endswith(".weight_orig") or key == "weight_orig":
            for leaf in ("weight_u", "weight_v"):
                k = key[:-len("weight_orig")] + leaf
                sd[k] = sd[k] / sd[k].norm()
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


def record_stats(kind, name, **kv):
    """Observed parity statistics, appended to $VCVITS_PARITY_STATS when set (the bounds in the tests are re-based on
    these files: profiles/r3_parity_stats_*.txt)."""
    path = os.environ.get("VCVITS_PARITY_STATS")
    if path:
        with open(path, "a") as f:
            f.write("%s %s %s\n" % (kind, name, " ".join("%s=%.4g" % (k, v) for k, v in kv.items())))


def close_kinked(name, a, b, tol=3e-4, tol_l2=5e-3, frac=0.01, frac4=0.005, cap=0.05, floor=0.0):
    """Comparison for gradients that passed through (leaky-)ReLU layers at full size.

    A pre-activation within fp32 rounding of zero can land on different sides of the kink on the two machines; that
    flips ONE derivative (1 vs slope) and moves every gradient element depending on it -- a cone of input positions, a
    row of a weight gradient, a little of every bias sum -- by up to a percent.  With ~1e8 activations per pass a
    handful of such flips is certain, so whole-model gradients cannot meet a 1e-4 max-norm bound element for element
    (the CPU reference run twice in different summation orders would not either).  The statistic used instead, re-based
    in round 3 on what the full-width tests observe on the MI355X (profiles/r3_parity_stats_f32.txt, two runs with
    different kernel sets: tensors of >= 65536 elements: worst relative L2 4.1e-3, worst share of elements beyond tol
    0.64 %, beyond 4 tol 0.12 %, worst single element 3.4 % of the scale; 4096 .. 65535 elements -- where ONE flip's cone is
    a visible share of the tensor -- up to 1.55 % beyond tol and 0.45 % beyond 4 tol, worst element 0.46 %; small tensors:
    worst element 1.1 % of the scale):
      * relative L2 error of the tensor <= tol_l2 (5e-3),
      * all but `frac` (1 %: the 0.64 % above with 1.5x margin -- which flips occur depends on the kernels' summation
        order, and the driver runs this suite with `-x`; 3 % under 65536 elements) of the elements within tol * max|b|
        (+ floor) and all but `frac4` (0.5 %; 1 %) within 4 tol * max|b|; tensors under 4096 elements (bias sums, which collect a little of EVERY
        flip) are held to 10 tol * max|b| instead, two elements excepted,
      * no element further than `cap` (5 %) of max|b| (+ floor) away: a wrong edge tile -- a few percent of a tensor off
        by the size of the values -- fails this and the L2 bound.
    The same layer shapes are compared strictly -- as linear launches without activations -- in
    tests/test_48k_gpu.py::test_period_conv_layers_strict and tests/test_conv_gpu.py."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs()
    assert bool(torch.isfinite(err).all()), name
    mx = b.abs().max().item()
    l2 = err.norm().item()
    slack = floor + 2e-6 * mx
    record_stats("kinked", name, n=err.numel(), rel_l2=l2 / (b.norm().item() + 1e-300),
                 frac_beyond_tol=float((err > tol * mx + slack).double().mean()),
                 frac_beyond_4tol=float((err > 4 * tol * mx + slack).double().mean()),
                 frac_beyond_10tol=float((err > 10 * tol * mx + slack).double().mean()),
                 max_err_over_scale=err.max().item() / (mx + floor + 1e-300))
    assert l2 <= tol_l2 * b.norm().item() + floor * err.numel() ** 0.5, "%s: relative L2 error %.3e" % (name, l2 / (b.norm().item() + 1e-30))
    assert err.max().item() <= cap * mx + floor, "%s: max err %.3e vs scale %.3e" % (name, err.max().item(), mx)
    if err.numel() < 4096:
        bad = int((err > 10 * tol * mx + slack).sum())  # a flip right under a bias moves that one sum
        assert bad <= 2, "%s: %d of %d elements off by more than %.1e of the scale" % (name, bad, err.numel(), 10 * tol)
        return
    if err.numel() < 65536:
        frac, frac4 = max(frac, 0.03), max(frac4, 0.01)
    bad = int((err > tol * mx + slack).sum())
    assert bad <= frac * err.numel() + 1, "%s: %d of %d elements off by more than %.1e of the scale" % (name, bad, err.numel(), tol)
    bad4 = int((err > 4 * tol * mx + slack).sum())
    assert bad4 <= frac4 * err.numel() + 1, "%s: %d of %d elements off by more than %.1e of the scale" % (name, bad4, err.numel(), 4 * tol)


def rank_against_f64(name, grads, ref32, ref64):
    """Ranking against float64 (verdict r5 #6): is the HIP step any further from the truth than torch-CPU fp32 is?

    Per tensor and per network: relative L2 distance to the float64 oracle step of (a) the HIP step, (b) the torch-CPU fp32
    oracle step (tensors whose gradient is analytically ~0 get an absolute floor from their network's largest gradient).
    What the MI355X shows (profiles/r6_f64_ranking.txt: 2,722 tensors of four full-width B = 2 batches and the headline
    B = 16 batch; profiles/r6_f64_rank_arithmetics.txt: the same step in the library's three fp32 arithmetics):
      * at B = 2 the two fp32 steps are statistically the SAME distance from float64 -- median hip / cpu32 ratio 0.44 .. 2.2
        per batch; each side has whole sub-networks at ~2e-4 where the other sits at 3e-7 (ONE leaky-ReLU kink flipped on that
        side: a pre-activation within fp32 rounding of zero), worst tensor 4.4e-3 (HIP) / 3.5e-3 (CPU fp32), worst network
        1.1e-4 / 1.1e-4; "HIP <= 2 x CPU per tensor" fails in BOTH directions (CPU fp32 against HIP on 43 .. 374 tensors per
        batch, HIP against CPU on 35 .. 256) -- no pair of fp32 implementations can hold it;
      * the headline batch in all arithmetics (tools/f64_rank.py): the generator's whole gradient is, at B = 2, 4.1e-5 from
        float64 with the six-term split kernels, 4.0e-5 with nine terms (exact operands), 3.9e-5 with the fmaf-chain
        kernels and 5.3e-5 on the CPU fp32 step; at B = 16: 9.8e-5 / 2.6e-5 / 3.1e-5 / 1.1e-5.  Every fp32 implementation
        lands somewhere between 1e-5 and 1e-4, by which kinks its summation order happens to flip -- bit-exact fp32
        arithmetic (the fmaf chain) included; the discriminators' gradients (shorter chains) sit at 1e-7 .. 7e-6 for all.
    So the assertions are absolute, with the CPU fp32 step's own distance recorded beside every number:
      * every tensor within 1e-2 of float64 (relative L2; observed worst 4.4e-3 HIP / 3.5e-3 CPU fp32) -- a wrong tile or a
        dropped term is 1e-1 and up;
      * every network's whole gradient within 5e-4 (observed worst 1.1e-4 on both sides);
      * the median tensor within 5e-4.
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
    med = statistics.median(eh for _, _, eh, _ in rows)
    assert med <= 5e-4, (name, med, statistics.median(ec for _, _, _, ec in rows))
    return rows, nets
