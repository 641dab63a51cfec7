"""GPU: packed copies of plain conv weights that live in an optimizer's flat parameter buffer are cached per region
(ops.register_param_region) and re-made in one batched launch after the region is written.  What must never happen is a
stale pack: an in-place write through torch (version bump) and a raw write followed by invalidate_weights both have to be
seen by the next launch, in the fp32 (split-operand) and bf16 kernels."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_region_packs_follow_the_weights(gpu, dtype):
    from vcvits_amd import ops
    g0 = torch.Generator().manual_seed(7)
    flat = torch.randn(2 * 64 * 64 * 3 + 64, generator=g0).to(gpu) * 0.1
    w1 = flat[:64 * 64 * 3].view(64, 64, 3)
    w2 = flat[64 * 64 * 3:2 * 64 * 64 * 3].view(64, 64, 3)
    x = torch.randn(2, 64, 300, generator=g0).to(gpu)
    ops.set_compute_dtype(dtype)
    ops.register_param_region(flat)
    try:
        def fwd():
            with torch.no_grad():
                return ops.conv1d(x, w1, None, pad=1), ops.conv1d(x, w2, None, pad=1)
        a1, a2 = fwd()
        b1, b2 = fwd()                      # second use: cached packs
        assert torch.equal(a1, b1) and torch.equal(a2, b2)
        with torch.no_grad():
            w1.mul_(2.0)                    # in-place write through torch: the version counter moves
        c1, c2 = fwd()
        assert torch.allclose(c1, 2.0 * a1, rtol=1e-5, atol=1e-6) and torch.allclose(c2, a2, rtol=1e-5, atol=1e-6)
        flat[64 * 64 * 3:2 * 64 * 64 * 3].mul_(-1.0)   # a write that w2's own version counter may not see ...
        ops.invalidate_weights(flat.data_ptr(), flat.data_ptr() + 4 * flat.numel())  # ... announced as the optimizer does
        d1, d2 = fwd()                      # first use after the invalidation: the recorded jobs are replayed in one launch
        assert torch.allclose(d1, c1, rtol=1e-5, atol=1e-6) and torch.allclose(d2, -a2, rtol=1e-5, atol=1e-6)
        e1, e2 = fwd()
        assert torch.equal(e1, d1) and torch.equal(e2, d2)
    finally:
        ops.unregister_param_region(flat)
        ops.invalidate_weights()
        ops.set_compute_dtype("f32")
