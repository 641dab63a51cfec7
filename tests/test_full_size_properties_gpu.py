"""GPU, BASELINE.json full sizes (configs/base.json widths, batch 16): size-independent properties of
the HIP path where an element-wise CPU oracle would take minutes -- flow round trip, linearity of the
MFMA conv family at the dominant layer shapes, stacked-vs-split discriminator equivalence, STFT
energy / peak location, generator range and determinism."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def test_flow_roundtrip_base_width(gpu):
    from vcvits_amd.model.flow import ResidualCouplingBlock
    torch.manual_seed(0)
    flow = ResidualCouplingBlock(256, 256, 5, 1, 4, gin_channels=256).to(gpu)
    with torch.no_grad():
        for n, p in flow.named_parameters():
            if ".post." in n:
                p.normal_(0.0, 0.02)  # zero-initialised in the reference (modules.py:314-315)
    B, T = 16, 384
    z = torch.randn(B, 256, T, device=gpu)
    g = torch.randn(B, 256, 1, device=gpu)
    lengths = torch.tensor([384] * 8 + [300] * 8, device=gpu)
    mask = (torch.arange(T, device=gpu)[None] < lengths[:, None]).unsqueeze(1).float()
    with torch.no_grad():
        z_p = flow(z, mask, g=g)
        z_back = flow(z_p, mask, g=g, reverse=True)
    assert z_p.shape == z.shape
    assert rel(z_back, z * mask) < 1e-4
    assert float((z_p - z * mask).abs().max()) > 1e-2  # the flow actually transformed something


@pytest.mark.parametrize("shape", [
    # B, C, M, H, P, K, stride, pad  -- the dominant period-discriminator layers at bench size
    (32, 1024, 1024, 102, 2, 5, 1, 2),
    (32, 512, 1024, 304, 2, 5, 3, 2),
    (32, 1024, 1024, 6, 37, 5, 1, 2),
])
def test_conv_linearity_and_gradient_adjointness(gpu, shape):
    """conv(a x1 + b x2) = a conv(x1) + b conv(x2), and <conv(x), r> = <x, dgrad(r)> = <w, wgrad(r, x)>."""
    from vcvits_amd import ops
    B, C, M, H, P, K, s, pad = shape
    torch.manual_seed(1)
    w = torch.randn(M, C, K, device=gpu) / math.sqrt(C * K)
    x1 = torch.randn(B, C, H, P, device=gpu)
    x2 = torch.randn(B, C, H, P, device=gpu)
    y1 = ops.conv_forward(x1, w, stride=s, pad=pad)
    y2 = ops.conv_forward(x2, w, stride=s, pad=pad)
    y12 = ops.conv_forward(0.5 * x1 - 2.0 * x2, w, stride=s, pad=pad)
    assert rel(y12, 0.5 * y1 - 2.0 * y2) < 2e-5
    r = torch.randn_like(y1)
    dx = ops.conv_dgrad(r, w, x1.shape, stride=s, pad=pad)
    dw = ops.conv_wgrad(r, x1, w.shape, stride=s, pad=pad)
    lhs = (y1.double() * r.double()).sum().item()
    assert abs(lhs - (x1.double() * dx.double()).sum().item()) < 1e-4 * abs(lhs) + 1e-2
    assert abs(lhs - (w.double() * dw.double()).sum().item()) < 1e-4 * abs(lhs) + 1e-2


def test_discriminator_stacked_equals_split(gpu):
    """The stacked (2B) pass used in both optimizer passes gives each signal exactly what a separate
    pass gives it (no cross-batch leakage), at full width and segment length."""
    from vcvits_amd.model.discriminators.discriminator import DiscriminatorP, DiscriminatorS
    torch.manual_seed(2)
    y = torch.rand(4, 1, 16384, device=gpu) * 1.8 - 0.9
    y_hat = torch.rand(4, 1, 16384, device=gpu) * 1.8 - 0.9
    for d in (DiscriminatorP(3).to(gpu), DiscriminatorP(37).to(gpu), DiscriminatorS().to(gpu)):
        with torch.no_grad():
            o_r, f_r = d(y)
            o_g, f_g = d(y_hat)
            o_c, f_c = d(torch.cat([y, y_hat]))
        assert rel(o_c[:4], o_r) < 1e-5 and rel(o_c[4:], o_g) < 1e-5
        for a, b, c in zip(f_r, f_g, f_c):
            assert rel(c[:4], a) < 1e-5 and rel(c[4:], b) < 1e-5


def test_stft_energy_and_peak(gpu):
    from vcvits_amd import mel_processing
    sr, n_fft, hop = 48000, 2048, 512
    t = torch.arange(196608, device=gpu, dtype=torch.float32)
    k0 = 200  # exactly on bin 200
    y = 0.5 * torch.sin(2 * math.pi * k0 * t / n_fft).unsqueeze(0).repeat(16, 1)
    spec = mel_processing.spectrogram_torch_audio(y, n_fft, sr, hop, n_fft)
    assert spec.shape == (16, 1025, 384)
    inner = spec[:, :, 4:-4]  # frames untouched by the zero padding
    assert int(inner.mean(dim=(0, 2)).argmax()) == k0
    # Hann window: peak magnitude = A * N / 4, neighbours half of it
    peak = inner[:, k0].mean().item()
    assert abs(peak - 0.5 * n_fft / 4) < 1e-3 * peak
    assert abs(inner[:, k0 + 1].mean().item() - peak / 2) < 2e-3 * peak
    assert float(inner[:, k0 + 3:].max()) < 1e-2 * peak + 2e-3


def test_generator_range_shape_determinism(gpu):
    from vcvits_amd.model.generator import Generator
    torch.manual_seed(3)
    gen = Generator(256, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 4, 2], 512, [16, 16, 4, 4]).to(gpu)
    z = torch.randn(16, 256, 32, device=gpu)
    with torch.no_grad():
        a = gen(z)
        b = gen(z)
    assert a.shape == (16, 1, 16384)
    assert float(a.abs().max()) <= 1.0
    assert torch.equal(a, b)  # forward kernels are deterministic (atomics are only used in weight gradients)
