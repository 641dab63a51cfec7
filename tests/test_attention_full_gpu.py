"""Full-width relative-position attention against the reference (tests/golden/attention_full.npz, generated from the
reference's own MultiHeadAttention by tools/make_goldens_attention_full.py: vits/model/transformer/
relative_attention_transformer.py:103-251) and against the oracle on the same inputs.

d_k in {64 (base: 256 channels / 4 heads), 32 (48k: 128 / 4)} x T in {204, 256, 500} with ragged lengths: the shapes the
fused kernels of csrc/attention.hip (`vcv_rel_attn_fwd` / `vcv_rel_attn_bwd2`) are built for -- the small fixture
attention.npz (d_k = 8) never reaches them.  GPU test: the HIP module's output, probabilities, input gradient and every
parameter gradient (i) element for element against the oracle run here on the CPU and (ii) at the fixture's 256 sampled
positions + float64 sums against the reference's values.  CPU test: the oracle against the fixture (pins the oracle at
full width on every box)."""
import numpy as np
import pytest
import torch

from golden_util import fill_state_dict, keys_shapes_of, load, rng_tensor

TOL = 1e-4  # north_star: "within 1e-4 fp32"
B = 3


def _case(g, ci):
    C, H, T = (int(v) for v in g["cases"][ci])
    seed = int(g["c%d_seed" % ci])
    rng = np.random.default_rng([seed, C, T])
    x = rng_tensor(rng, (B, C, T))
    r = rng_tensor(rng, (B, C, T))
    lengths = torch.from_numpy(g["c%d_lengths" % ci])
    return C, H, T, seed, x, r, lengths


def _against_fixture(g, ci, name, t):
    tag = "c%d_%s" % (ci, name)
    if tag + "_idx" not in g.files:
        return
    flat = t.detach().double().cpu().reshape(-1)
    idx = torch.from_numpy(g[tag + "_idx"])
    want = torch.from_numpy(g[tag + "_vals"])
    scale = float(g[tag + "_max"])
    err = (flat[idx] - want).abs().max().item()
    assert err <= TOL * scale + 2e-6, "%s: sampled elements off by %.3e (scale %.3e)" % (tag, err, scale)
    sums = g[tag + "_sums"]
    # sums over up to 3e6 elements: float64 accumulation of fp32 values; bound = tol x the abs-sum
    assert abs(flat.sum().item() - sums[0]) <= TOL * sums[1] + 1e-5, (tag, flat.sum().item(), sums[0])
    assert abs(flat.abs().sum().item() - sums[1]) <= TOL * sums[1] + 1e-5, (tag, flat.abs().sum().item(), sums[1])


def _oracle(C, H, T, seed, x, r, lengths, module_keys):
    from oracle import vits_oracle as O
    sd = fill_state_dict(module_keys, seed)
    leaves = {"a." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    xm = O.sequence_mask(lengths, T).unsqueeze(1).float()
    am = xm.unsqueeze(2) * xm.unsqueeze(-1)
    y, p = O.rel_attention(leaves, "a", xo, am, H, 4)
    names = [k for k, _ in module_keys]
    grads = torch.autograd.grad((y * r).sum(), [xo] + [leaves["a." + n] for n in names], allow_unused=True)
    return y.detach(), p.detach(), grads[0], dict(zip(names, grads[1:])), sd


MODULE_KEYS = lambda C, H: [("conv_q.weight", (C, C, 1)), ("conv_q.bias", (C,)), ("conv_k.weight", (C, C, 1)), ("conv_k.bias", (C,)),  # noqa: E731
                            ("conv_v.weight", (C, C, 1)), ("conv_v.bias", (C,)), ("conv_o.weight", (C, C, 1)), ("conv_o.bias", (C,)),
                            ("emb_rel_k", (1, 9, C // H)), ("emb_rel_v", (1, 9, C // H))]


@pytest.mark.parametrize("ci", range(6))
def test_oracle_attention_full_width_vs_reference_fixture(ci):
    g = load("attention_full.npz")
    C, H, T, seed, x, r, lengths = _case(g, ci)
    torch.set_num_threads(4)
    y, p, dx, dps, _ = _oracle(C, H, T, seed, x, r, lengths, MODULE_KEYS(C, H))
    _against_fixture(g, ci, "y", y)
    _against_fixture(g, ci, "attn", p)
    _against_fixture(g, ci, "dx", dx)
    for n, d in dps.items():
        if d is not None and n != "conv_k.bias":  # (analytically zero: rounding noise on both sides, see the GPU test)
            _against_fixture(g, ci, "dp_" + n, d)


def _close(name, a, b, tol=TOL, atol=2e-6):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    bound = tol * b.abs().max().item() + atol
    assert err <= bound, "%s: abs err %.3e > %.3e" % (name, err, bound)


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(6))
def test_hip_attention_full_width_vs_reference_and_oracle(gpu, ci):
    from vcvits_amd import ops
    from vcvits_amd.model.transformer.relative_attention_transformer import MultiHeadAttention
    g = load("attention_full.npz")
    C, H, T, seed, x, r, lengths = _case(g, ci)
    m = MultiHeadAttention(C, C, H, p_dropout=0.0, window_size=4).eval()
    keys = keys_shapes_of(m)
    assert [k for k, _ in keys] == [k for k, _ in MODULE_KEYS(C, H)] or set(k for k, _ in keys) == set(k for k, _ in MODULE_KEYS(C, H))
    y_o, p_o, dx_o, dps_o, sd = _oracle(C, H, T, seed, x, r, lengths, keys)
    m.load_state_dict(sd)
    m = m.to(gpu)
    m.store_attn = True
    xg = x.to(gpu).requires_grad_(True)
    ar = torch.arange(T, device=gpu)
    xm = (ar.unsqueeze(0) < lengths.to(gpu).unsqueeze(1)).unsqueeze(1).float()
    before = ops.LAUNCH_COUNTS["attn_fused"]
    y = m(xg, xg, x_mask=xm)
    fused = ops.LAUNCH_COUNTS["attn_fused"] - before
    assert fused == 1, "C=%d T=%d did not run the fused attention launch" % (C, T)
    (y * r.to(gpu)).sum().backward()
    # (i) element for element against the oracle
    _close("y", y, y_o)
    _close("attn", m.attn, p_o)
    _close("dx", xg.grad, dx_o)
    qb = float(dps_o["conv_q.bias"].abs().max())
    for n, p in m.named_parameters():
        if dps_o.get(n) is not None:
            assert p.grad is not None, n
            if n == "conv_k.bias":
                # analytically ZERO (a constant added to every key's logit cancels in the softmax): both sides hold fp32
                # rounding noise of the row sums, ~1e-6 against a query-bias gradient of order 1
                assert float(p.grad.abs().max()) <= 1e-4 * qb + 2e-5 and float(dps_o[n].abs().max()) <= 1e-4 * qb + 2e-5
                continue
            _close("dp_" + n, p.grad, dps_o[n], tol=2e-4 if n.startswith("emb_rel") else TOL, atol=5e-6)
    # (ii) against the values the reference's own module produced
    _against_fixture(g, ci, "y", y)
    _against_fixture(g, ci, "attn", m.attn)
    _against_fixture(g, ci, "dx", xg.grad)
    for n, p in m.named_parameters():
        if p.grad is not None and not n.startswith("emb_rel") and n != "conv_k.bias":
            _against_fixture(g, ci, "dp_" + n, p.grad)
