"""GPU: the fused attention kernels (attention.hip: one launch forward, two backward) against the unfused path they
replace (grouped GEMM launches + softmax kernels, itself pinned by the `attention.npz` / `transformer.npz` goldens of
relative_attention_transformer.py:150-182) at the real widths of both configs, ragged masks, with and without dropout
(same counter-based masks on both paths), and in bf16 mode against fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [  # B, H, dk, T, window, pdrop
    (4, 4, 64, 204, 4, 0.0),   # base: hidden 256, 4 heads
    (4, 4, 64, 204, 4, 0.1),
    (3, 4, 32, 160, 4, 0.1),   # 48k: hidden 128
    (2, 2, 8, 37, 4, 0.0),     # the goldens' width, one ragged tile
    (2, 4, 64, 333, 4, 0.1),   # tile edges on both axes
    (1, 4, 32, 800, 4, 0.0),   # 16 s of content frames (inference); past 896 frames the unfused path runs
    # edges of the wave-per-query-tile forward (T <= 256, 32 / 64 channels per head): one partial tile, exact tile
    # multiples, all eight key tiles, idle waves in the last workgroup, narrow and wide windows
    (2, 2, 32, 31, 4, 0.0),
    (2, 2, 64, 32, 4, 0.1),
    (2, 1, 32, 33, 2, 0.0),
    (2, 2, 64, 225, 7, 0.1),
    (1, 2, 32, 256, 4, 0.0),
    (2, 2, 64, 129, 1, 0.1),
]


def _run(ops, fused, q, k, v, ek, ev, mask, H, w, p, gy, seed_state):
    ops._ATTN_FUSED[0] = fused
    ops.set_seed_state(seed_state)
    ts = [t.clone().requires_grad_(True) for t in (q, k, v, ek, ev)]
    before = ops.LAUNCH_COUNTS["attn_fused"]
    out, attn = ops.rel_attention(ts[0], ts[1], ts[2], ts[3], ts[4], mask, H, w, p, training=True, want_attn=True)
    assert (ops.LAUNCH_COUNTS["attn_fused"] - before) == (1 if fused else 0)
    out.backward(gy)
    return [out.detach(), attn.detach()] + [t.grad.detach() for t in ts]


def rel(a, b):
    return (a.double() - b.double()).abs().max().item() / (b.double().abs().max().item() + 1e-30)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d-H%d-dk%d-T%d-p%g" % (c[0], c[1], c[2], c[3], c[5]))
def test_fused_attention_matches_unfused(gpu, case):
    from vcvits_amd import ops
    B, H, dk, T, w, p = case
    rng = np.random.default_rng(100 + T)
    t = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)).to(gpu)
    q, k, v, gy = t(B, H * dk, T), t(B, H * dk, T), t(B, H * dk, T), t(B, H * dk, T)
    ek, ev = t(1, 2 * w + 1, dk) * dk ** -0.5, t(1, 2 * w + 1, dk) * dk ** -0.5
    mask = torch.ones(B, T, device=gpu)
    if B > 1:
        mask[1, T - T // 5:] = 0.0
        if T > 40:
            mask[0, T // 3] = 0.0  # an interior hole: a masked query row and a masked key column
    try:
        a = _run(ops, True, q, k, v, ek, ev, mask, H, w, p, gy, 77)
        b = _run(ops, False, q, k, v, ek, ev, mask, H, w, p, gy, 77)
        names = ("out", "attn", "dq", "dk", "dv", "dembk", "dembv")
        for n, x, y in zip(names, a, b):
            assert torch.isfinite(x).all(), n
            assert rel(x, y) < 2e-5, (n, rel(x, y))
        if p > 0:  # the dropout really dropped (and the two paths drew the same mask: attn agreed above)
            frac = float((a[1] == 0).float().mean())
            assert 0.5 * p < frac < 1.0, frac
        # bf16 mode: operands rounded on the way into the matrix cores, fp32 softmax / accumulate
        ops.set_compute_dtype("bf16")
        c = _run(ops, True, q, k, v, ek, ev, mask, H, w, p, gy, 77)
        for n, x, y in zip(names, c, a):
            if n == "attn" and p > 0:
                continue  # a probability at the dropout threshold... the mask is the same, the values differ by rounding
            assert rel(x, y) < 3e-2, ("bf16", n, rel(x, y))
        assert rel(c[0], a[0]) > 1e-6  # it is a different arithmetic
    finally:
        ops._ATTN_FUSED[0] = True
        ops.set_compute_dtype("f32")


def test_attn_is_written_only_on_request(gpu):
    from vcvits_amd import ops
    B, H, dk, T, w = 2, 4, 64, 204, 4
    g = torch.Generator().manual_seed(3)
    t = lambda *s: torch.randn(*s, generator=g).to(gpu)
    q, k, v = t(B, H * dk, T), t(B, H * dk, T), t(B, H * dk, T)
    ek, ev = t(1, 9, dk), t(1, 9, dk)
    mask = torch.ones(B, T, device=gpu)
    with torch.no_grad():
        o1, a1 = ops.rel_attention(q, k, v, ek, ev, mask, H, w, want_attn=True)
        torch.cuda.synchronize()
        m0 = torch.cuda.memory_allocated()
        o2, a2 = ops.rel_attention(q, k, v, ek, ev, mask, H, w, want_attn=False)
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - m0
    assert a2 is None and a1.shape == (B, H, T, T)
    assert torch.equal(o1, o2)
    assert grown <= o2.numel() * 4 + 4096, "an eval-mode call without `attn` allocated a [B,H,T,T] buffer"
