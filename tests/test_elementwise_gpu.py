"""GPU parity of the streaming kernels, STFT, mel and AdamW against torch CPU ops (fp32)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cmp(name, got, ref, tol=2e-5):
    got = got.detach().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item() / scale
    assert err <= tol, "%s: rel err %.3e" % (name, err)


@pytest.mark.parametrize("shape", [(32, 1, 5, 1), (1024, 1024, 5), (16, 1, 15), (512, 256, 16), (1024, 4, 41)])
def test_weight_norm(gpu, shape):
    from vcvits_amd import ops
    gen = torch.Generator().manual_seed(1)
    v = torch.randn(shape, generator=gen)
    g = torch.rand((shape[0],) + (1,) * (len(shape) - 1), generator=gen) + 0.5
    vr, gr = v.clone().requires_grad_(True), g.clone().requires_grad_(True)
    wr = torch._weight_norm(vr, gr, 0)
    gw = torch.randn(shape, generator=gen)
    wr.backward(gw)
    vg, gg = v.clone().to(gpu).requires_grad_(True), g.clone().to(gpu).requires_grad_(True)
    wg = ops.weight_norm(vg, gg)
    wg.backward(gw.to(gpu))
    _cmp("w", wg, wr.detach())
    _cmp("dv", vg.grad, vr.grad)
    _cmp("dg", gg.grad, gr.grad)


def test_weight_norm_many(gpu):
    """One launch for a list of (v, g) pairs == torch._weight_norm per pair, forward and gradients; a second
    call (cached descriptor table) with updated values follows the new values."""
    from vcvits_amd import ops
    shapes = [(32, 1, 5, 1), (1024, 1024, 5), (16, 1, 15), (512, 256, 16), (1, 1024, 3), (1024, 4, 41)]
    gen = torch.Generator().manual_seed(11)
    vs = [torch.randn(s, generator=gen) for s in shapes]
    gs = [torch.rand((s[0],) + (1,) * (len(s) - 1), generator=gen) + 0.5 for s in shapes]
    gws = [torch.randn(s, generator=gen) for s in shapes]
    vg = [v.clone().to(gpu).requires_grad_(True) for v in vs]
    gg = [g.clone().to(gpu).requires_grad_(True) for g in gs]
    for rnd in range(2):
        if rnd == 1:
            with torch.no_grad():
                for i in range(len(vs)):
                    vs[i] = vs[i] * 0.5 + 0.1
                    vg[i].copy_(vs[i])
                    vg[i].grad = gg[i].grad = None
        ws = ops.weight_norm_many(vg, gg)
        torch.autograd.backward(list(ws), [t.to(gpu) for t in gws])
        for i, s in enumerate(shapes):
            vr, gr = vs[i].clone().requires_grad_(True), gs[i].clone().requires_grad_(True)
            wr = torch._weight_norm(vr, gr, 0)
            wr.backward(gws[i])
            _cmp("w%d" % i, ws[i], wr.detach())
            _cmp("dv%d" % i, vg[i].grad, vr.grad)
            _cmp("dg%d" % i, gg[i].grad, gr.grad)


def test_avg3_pad_pool(gpu):
    from vcvits_amd import ops
    gen = torch.Generator().manual_seed(2)
    a, b, c = (torch.randn(2, 8, 100, generator=gen) for _ in range(3))
    leaves = [t.clone().requires_grad_(True) for t in (a, b, c)]
    ref = (leaves[0] + leaves[1] + leaves[2]) / 3
    gy = torch.randn(ref.shape, generator=gen)
    ref.backward(gy)
    gl = [t.clone().to(gpu).requires_grad_(True) for t in (a, b, c)]
    out = ops.avg3(*gl)
    out.backward(gy.to(gpu))
    _cmp("avg3", out, ref.detach())
    _cmp("davg3", gl[1].grad, leaves[1].grad)

    for T, period in [(16384, 3), (16384, 37), (8193, 7), (100, 23)]:
        x = torch.randn(3, 1, T, generator=gen)
        n_pad = period - T % period
        xr = x.clone().requires_grad_(True)
        yr = F.pad(xr, (0, n_pad), "reflect")
        gy = torch.randn(yr.shape, generator=gen)
        yr.backward(gy)
        xg = x.clone().to(gpu).requires_grad_(True)
        yg = ops.reflect_pad_right(xg, n_pad)
        yg.backward(gy.to(gpu))
        _cmp("pad", yg, yr.detach())
        _cmp("dpad", xg.grad, xr.grad)

    for T in [16384, 8193, 4097, 2049, 50]:
        x = torch.randn(3, 1, T, generator=gen)
        xr = x.clone().requires_grad_(True)
        yr = F.avg_pool1d(xr, 4, 2, 2)
        gy = torch.randn(yr.shape, generator=gen)
        yr.backward(gy)
        xg = x.clone().to(gpu).requires_grad_(True)
        yg = ops.avgpool4(xg)
        yg.backward(gy.to(gpu))
        _cmp("pool", yg, yr.detach())
        _cmp("dpool", xg.grad, xr.grad)


def test_losses(gpu):
    from vcvits_amd import ops
    gen = torch.Generator().manual_seed(3)
    a_list = [torch.randn(2, 4, n, generator=gen) for n in (10, 333, 5000)]
    b_list = [torch.randn(2, 4, n, generator=gen) for n in (10, 333, 5000)]
    ar = [a.clone().requires_grad_(True) for a in a_list]
    ref = sum(torch.mean(torch.abs(b - a)) for a, b in zip(ar, b_list)) * 2
    ref2 = sum(torch.mean((1 - a) ** 2) for a in ar)
    (ref * 3 + ref2).backward()
    ag = [a.clone().to(gpu).requires_grad_(True) for a in a_list]
    bg = [b.to(gpu) for b in b_list]
    out = ops.l1_mean_sum(ag, bg, weight=2.0)
    out2 = ops.sq_mean_sum(ag, 1.0)
    (out * 3 + out2).backward()
    _cmp("l1", out, ref.detach())
    _cmp("sq", out2, ref2.detach())
    for i in range(3):
        _cmp("dl%d" % i, ag[i].grad, ar[i].grad)


@pytest.mark.parametrize("reflect", [False, True])
@pytest.mark.parametrize("T", [16384, 5120])
def test_stft_mag(gpu, reflect, T):
    from vcvits_amd import ops
    gen = torch.Generator().manual_seed(4)
    y = torch.rand(3, T, generator=gen) * 1.8 - 0.9
    yr = y.clone().requires_grad_(True)
    win = torch.hann_window(2048)
    yp = F.pad(yr.unsqueeze(1), (768, 768), mode="reflect" if reflect else "constant").squeeze(1)
    spec = torch.stft(yp, 2048, hop_length=512, win_length=2048, window=win, center=False,
                      normalized=False, onesided=True, return_complex=True)
    mr = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-6)
    gm = torch.randn(mr.shape, generator=gen)
    mr.backward(gm)
    yg = y.clone().to(gpu).requires_grad_(True)
    mg = ops.stft_mag(yg, 2048, 512, 768, reflect, 1e-6)
    mg.backward(gm.to(gpu))
    _cmp("mag", mg, mr.detach(), tol=1e-5)
    _cmp("dy", yg.grad, yr.grad, tol=1e-4)


def test_mel_log(gpu):
    from vcvits_amd import ops
    gen = torch.Generator().manual_seed(5)
    spec = torch.rand(2, 1025, 40, generator=gen) * 3
    mel = torch.rand(256, 1025, generator=gen) * 0.01
    mel[:3] = 0  # empty filters -> clamp branch
    sr = spec.clone().requires_grad_(True)
    ref = torch.log(torch.clamp(torch.matmul(mel.unsqueeze(0), sr), min=1e-5))
    gy = torch.randn(ref.shape, generator=gen)
    ref.backward(gy)
    sg = spec.clone().to(gpu).requires_grad_(True)
    out = ops.mel_log(sg, mel.to(gpu), 1e-5)
    out.backward(gy.to(gpu))
    _cmp("mel", out, ref.detach())
    _cmp("dspec", sg.grad, sr.grad)


def test_adamw(gpu):
    from vcvits_amd import ops
    gen = torch.Generator().manual_seed(6)
    p0 = torch.randn(10000, generator=gen)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], 2e-4, betas=(0.8, 0.99), eps=1e-9)
    pg = p0.clone().to(gpu)
    m = torch.zeros_like(pg)
    v = torch.zeros_like(pg)
    for step in range(1, 4):
        g = torch.randn(10000, generator=gen)
        pr.grad = g.clone()
        opt.step()
        ops.adamw_step(pg, g.to(gpu), m, v, 2e-4, (0.8, 0.99), 1e-9, 0.01, step)
    _cmp("adamw", pg, pr.detach(), tol=1e-6)


@pytest.mark.parametrize("B,T,C,rows", [(16, 204, 128, 512), (32, 204, 256, 512), (3, 37, 40, 9), (64, 1, 256, 512)])
def test_embedding_t_forward_and_table_gradient(gpu, B, T, C, rows):
    """ops.embedding_t == F.embedding(idx, W).transpose(1, -1) (content_encoder.py:58-60; emb_g(sid).unsqueeze(-1) at T = 1),
    forward and the table gradient (one workgroup per row, fixed summation order: bit-reproducible), with and without a
    gradient sink; at sizes on both sides of torch's 3072-index switch to its sort-based backward."""
    import torch.nn.functional as F
    from vcvits_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    W = torch.randn(rows, C, generator=g).to(gpu).requires_grad_(True)
    idx = torch.randint(0, rows, (B, T), generator=g).to(gpu)
    r = torch.randn(B, C, T, generator=g).to(gpu)
    y = ops.embedding_t(idx, W)
    ref = F.embedding(idx.cpu(), W.detach().cpu()).transpose(1, -1)
    assert torch.equal(y.detach().cpu(), ref)
    (y * r).sum().backward()
    Wc = W.detach().cpu().double().requires_grad_(True)
    (F.embedding(idx.cpu(), Wc).transpose(1, -1) * r.cpu().double()).sum().backward()
    err = (W.grad.cpu().double() - Wc.grad).abs().max().item()
    assert err <= 1e-5 * Wc.grad.abs().max().item() + 1e-6, err
    g1 = W.grad.clone()
    W.grad = None
    (ops.embedding_t(idx, W) * r).sum().backward()
    assert torch.equal(g1, W.grad)  # fixed order: bit for bit
    # gradient sink: added onto the registered view, autograd receives nothing
    sink = torch.ones(rows, C, device=gpu)
    ops.register_grad_sink(W, sink)
    try:
        W.grad = None
        (ops.embedding_t(idx, W) * r).sum().backward()
        assert W.grad is None and torch.allclose(sink - 1.0, g1, rtol=1e-5, atol=1e-5)
    finally:
        ops.unregister_grad_sink(W)
    if T == 1:
        y1 = ops.embedding_t(idx[:, 0], W)
        assert torch.equal(y1, y.detach())
