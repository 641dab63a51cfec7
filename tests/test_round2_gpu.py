"""GPU: regression tests for the round-1 advisor findings -- ResGradLink at training widths (incl. a partial backward
through torch.autograd.grad), AdamW's treatment of parameters without a gradient, raw writes into the flat parameter
buffer invalidating cached derived weights, and the prior-sample kernel."""
import numpy as np
import pytest
import torch

from golden_util import fill_state_dict, keys_shapes_of

pytestmark = pytest.mark.gpu


def close(name, a, b, tol=1e-4):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    err = (a - b).abs().max().item()
    assert err <= tol * b.abs().max().item() + 2e-6, "%s: %.3e" % (name, err)


@pytest.mark.parametrize("k", [3, 11])
def test_resblock1_training_width_gradients(gpu, k):
    """C = 64 >= 32 and T = 300 > 64: the fused stride-1 flipped-weight data gradient with the residual gradient added in
    its epilogue (ResGradLink) -- full backward and a partial one (torch.autograd.grad w.r.t. x only)."""
    from oracle import vits_oracle as O
    from vcvits_amd.model.modules import ResBlock1
    rb = ResBlock1(64, k, (1, 3, 5))
    sd = fill_state_dict(keys_shapes_of(rb), 70 + k)
    rb.load_state_dict(sd)
    rb = rb.to(gpu)
    rng = np.random.default_rng(k)
    x = torch.from_numpy(rng.standard_normal((3, 64, 300)).astype(np.float32))
    r = torch.from_numpy(rng.standard_normal((3, 64, 300)).astype(np.float32))
    sdc = {"r." + n: v.clone().requires_grad_(True) for n, v in sd.items()}
    xc = x.clone().requires_grad_(True)
    yo = O.resblock1_forward(sdc, "r", xc, k)
    (yo * r).sum().backward()
    xg = x.to(gpu).requires_grad_(True)
    y = rb(xg)
    close("y", y, yo)
    (y * r.to(gpu)).sum().backward()
    close("dx", xg.grad, xc.grad, tol=2e-4)
    for n, p in rb.named_parameters():
        close("d" + n, p.grad, sdc["r." + n].grad, tol=3e-4)
    # partial backward: only dx requested
    xg2 = x.to(gpu).requires_grad_(True)
    (dx2,) = torch.autograd.grad((rb(xg2) * r.to(gpu)).sum(), [xg2])
    close("dx via autograd.grad", dx2, xc.grad, tol=2e-4)


def test_adamw_skips_parameters_without_gradient(gpu):
    """torch.optim.AdamW leaves a parameter whose .grad is None untouched (no decay, no moment decay, no step count);
    FlatAdamW tracks which parameters a backward pass reached and updates only those, with per-parameter steps."""
    from vcvits_amd.light.optim import FlatAdamW
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ropt = torch.optim.AdamW(ref, 1e-2, betas=(0.8, 0.99), eps=1e-9)
    gp = [torch.nn.Parameter(p.detach().clone().to(gpu)) for p in ps]
    opt = FlatAdamW(gp, 1e-2, betas=(0.8, 0.99), eps=1e-9)
    use = [(0, 1, 2), (0, 2), (0, 2), (0, 1, 2), (1,)]
    for step, idx in enumerate(use):
        ropt.zero_grad(set_to_none=True)
        opt.zero_grad()
        loss_r = sum((ref[i] ** 2).sum() * (i + 1 + step) for i in idx)
        loss_g = sum((gp[i] ** 2).sum() * (i + 1 + step) for i in idx)
        loss_r.backward()
        loss_g.backward()
        ropt.step()
        opt.step()
        for a, b in zip(gp, ref):
            close("param after step %d" % step, a, b, tol=1e-5)
    assert sorted(opt._pstep) == [3, 4, 4]


def test_raw_write_into_flat_buffer_invalidates_derived_weights(gpu):
    """a broadcast (or any raw write) into FlatAdamW.flat must drop cached weight-norm results / packs --
    torch's version counters of the per-parameter views do not move."""
    from vcvits_amd.light.optim import FlatAdamW
    from vcvits_amd.model.discriminators.discriminator import DiscriminatorS
    torch.manual_seed(0)
    d = DiscriminatorS().to(gpu)
    opt = FlatAdamW(d.parameters(), 1e-3)
    y = torch.randn(2, 1, 4096, device=gpu)
    with torch.no_grad():
        a, _ = d(y)
        a2, _ = d(y)
        close("same weights, same result", a, a2, tol=1e-6)
        opt.write_flat(lambda flat: flat.mul_(0.5))   # what dist.broadcast does: writes behind torch's back
        b, _ = d(y)
        ref = DiscriminatorS().to(gpu)
        ref.load_state_dict({k: v.clone() for k, v in d.state_dict().items()})
        c, _ = ref(y)
    assert not torch.allclose(a, b)
    close("forward after raw write", b, c, tol=1e-6)


def test_prior_sample(gpu):
    from vcvits_amd import ops
    m, l, n = (torch.randn(2, 8, 50, device=gpu) for _ in range(3))
    close("prior", ops.prior_sample(m, l, n, 0.667), m + n * torch.exp(l) * 0.667, tol=1e-6)


@pytest.mark.parametrize("shape", [(8, 1024, 1024, 16, 13, 5, 1), (8, 128, 512, 141, 13, 5, 3), (4, 64, 64, 4096, 1, 11, 1)])
def test_weight_gradient_is_bit_reproducible(gpu, shape):
    """the split (batch, position) reduction of the MFMA weight-gradient kernel is combined through
    per-workgroup slabs added in a fixed order (VcvWgradArgs.slab), not fp32 atomics: two identical launches give
    identical bits -- and the atomics mode still gives the same numbers up to summation order."""
    from vcvits_amd import ops
    B, C, M, T, P, K, s = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn((B, C, T, P) if P > 1 else (B, C, T), generator=g).to(gpu)
    Tout = (T + 4 - K) // s + 1 if P > 1 else T
    dy = torch.randn((B, M, Tout, P) if P > 1 else (B, M, Tout), generator=g).to(gpu)
    pad = 2 if P > 1 else (K - 1) // 2
    runs = []
    for det in (True, True, False):
        ops.set_deterministic(det)
        try:
            runs.append(ops.conv_wgrad(dy, x, (M, C, K), stride=s, pad=pad).clone())
        finally:
            ops.set_deterministic(True)
    assert torch.equal(runs[0], runs[1])
    close("atomics vs slabs", runs[2], runs[0], tol=2e-5)


def test_stft_backward_is_bit_reproducible_and_exact(gpu):
    """Zero-pad STFT backward: every sample is owned by one workgroup that gathers its overlapping frames (no atomics):
    identical bits on repeated runs; values against torch CPU autograd, including ragged lengths (tail ownership)."""
    import torch.nn.functional as F
    from vcvits_amd import ops
    for T in (16384, 9000, 2048 + 512 * 5 + 77):
        g = torch.Generator().manual_seed(T)
        y = torch.rand(3, T, generator=g) * 1.8 - 0.9
        r = None
        outs = []
        for rep in range(2):
            yg = y.to(gpu).requires_grad_(True)
            mg = ops.stft_mag(yg, 2048, 512, 768, False, 1e-6)
            if r is None:
                r = torch.randn(mg.shape, generator=g)
            (mg * r.to(gpu)).sum().backward()
            outs.append(yg.grad.clone())
        assert torch.equal(outs[0], outs[1])
        yc = y.clone().requires_grad_(True)
        spec = torch.stft(F.pad(yc, (768, 768)), 2048, hop_length=512, win_length=2048, window=torch.hann_window(2048),
                          center=False, normalized=False, onesided=True, return_complex=True)
        mag = torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-6)
        close("mag T=%d" % T, mg, mag, tol=1e-5)
        (mag * r).sum().backward()
        close("dy T=%d" % T, outs[0], yc.grad, tol=2e-5)
