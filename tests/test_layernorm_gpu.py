"""GPU: channel LayerNorm of (x + y) over C for [B, C, T] (modules.py LayerNorm as the transformer uses it, x + residual
folded in) against torch's layer_norm, forward and every gradient: the register-resident kernels for the two configs'
widths (C = 256 / 128), ragged time tiles, and the generic kernels (another width)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(32, 256, 204), (16, 128, 204), (3, 256, 37), (2, 128, 64), (2, 96, 50)],
                         ids=lambda s: "B%d-C%d-T%d" % s)
@pytest.mark.parametrize("with_y", [True, False])
def test_layernorm_c_matches_torch(gpu, shape, with_y):
    from vcvits_amd import ops
    B, C, T = shape
    g0 = torch.Generator().manual_seed(B * 1000 + C + T)
    r = lambda *s: torch.randn(*s, generator=g0).to(gpu)
    x, y = r(B, C, T).requires_grad_(True), (r(B, C, T).requires_grad_(True) if with_y else None)
    ga, be = r(C).requires_grad_(True), r(C).requires_grad_(True)
    gy = r(B, C, T)
    out = ops.layernorm_c(x, y, ga, be)
    out.backward(gy)
    x2, g2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, ga, be))
    y2 = y.detach().clone().requires_grad_(True) if with_y else None
    inp = x2 + y2 if with_y else x2
    ref = torch.nn.functional.layer_norm(inp.transpose(1, 2), (C,), g2, b2, 1e-5).transpose(1, 2)
    ref.backward(gy)
    pairs = [("out", out, ref), ("dx", x.grad, x2.grad), ("dgamma", ga.grad, g2.grad), ("dbeta", be.grad, b2.grad)]
    if with_y:
        pairs.append(("dy", y.grad, y2.grad))
    for n, a, b in pairs:
        err = float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30))
        assert err < 2e-5, (n, err)
