"""STFT sizes other than the reference configs' 2048 / 512 / 2048 (vits/mel_processing.py:54-96 take n_fft, hop_size and
win_size as arguments; torch.stft pads a shorter window to n_fft).  Vectors: tests/golden/stft_sizes.npz, produced by the
reference's spectrogram_torch (tools/make_goldens_stft_sizes.py).

CPU: the oracle against those vectors.  GPU (-m gpu): mel_processing on the HIP kernels against them (n_fft = 2048 on the
tuned kernels with a shorter window or another hop, other powers of two on the generic radix-2 kernels, other even sizes
on the direct DFT), the zero-pad variant
and the input gradient against the oracle, the source-audio pipeline (complex STFT -> inverse STFT) at the same sizes."""
import numpy as np
import pytest
import torch

from golden_util import load
from oracle import vits_oracle as O


def _sizes(g):
    return [tuple(int(v) for v in row) for row in g["sizes"]]


def close(name, a, b, tol, atol=1e-6):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= tol * b.abs().max().item() + atol, "%s: abs err %.3e (scale %.3e)" % (name, err, b.abs().max().item())


def test_oracle_vs_reference_vectors():
    g = load("stft_sizes.npz")
    y = torch.from_numpy(g["y"])
    for n_fft, hop, win in _sizes(g):
        close("%d/%d/%d" % (n_fft, hop, win), O.spectrogram(y, n_fft, hop, win, reflect=True), g["spec_%d_%d_%d" % (n_fft, hop, win)],
              tol=2e-5)


@pytest.mark.gpu
def test_spectrogram_torch_at_other_sizes(gpu):
    from vcvits_amd import mel_processing
    g = load("stft_sizes.npz")
    y = torch.from_numpy(g["y"]).to(gpu)
    for n_fft, hop, win in _sizes(g):
        spec = mel_processing.spectrogram_torch(y, n_fft, 22050, hop, win, center=False)
        close("%d/%d/%d" % (n_fft, hop, win), spec, g["spec_%d_%d_%d" % (n_fft, hop, win)], tol=1e-5)  # north_star: 1e-5 on STFT


@pytest.mark.gpu
@pytest.mark.parametrize("reflect", [True, False])
def test_gradients_and_zero_pad_at_other_sizes(gpu, reflect):
    from vcvits_amd import mel_processing
    rng = np.random.default_rng(3)
    fn = mel_processing.spectrogram_torch if reflect else mel_processing.spectrogram_torch_audio
    for n_fft, hop, win, T in ((1024, 256, 1024, 4096), (512, 128, 400, 3000), (4096, 1024, 4096, 9000), (2048, 512, 1200, 6000),
                               (128, 32, 128, 777), (1280, 320, 1280, 5000), (400, 160, 320, 2000), (3000, 750, 3000, 8000)):
        y = torch.from_numpy((rng.standard_normal((3, T)) * 0.3).astype(np.float32))
        yc = y.clone().requires_grad_(True)
        ref = O.spectrogram(yc, n_fft, hop, win, reflect=reflect)
        r = torch.from_numpy(rng.standard_normal(tuple(ref.shape)).astype(np.float32))
        (ref * r).sum().backward()
        yg = y.to(gpu).requires_grad_(True)
        out = fn(yg, n_fft, 22050, hop, win, center=False)
        (out * r.to(gpu)).sum().backward()
        tag = "%d/%d/%d %s" % (n_fft, hop, win, "reflect" if reflect else "zero")
        close("spec " + tag, out, ref.detach(), tol=1e-5)
        close("dy " + tag, yg.grad, yc.grad, tol=1e-4)
    # mel of another size: the projection takes the bin count from the filterbank
    mel = mel_processing.mel_spectrogram_torch(y.to(gpu), 1024, 80, 22050, 256, 1024, 0, None)
    sp = O.spectrogram(y, 1024, 256, 1024, reflect=True)
    ref = O.spec_to_mel(sp, torch.from_numpy(O.mel_filterbank(22050, 1024, 80, 0, None)))
    close("mel 1024", mel, ref, tol=1e-4)


@pytest.mark.gpu
def test_unsupported_sizes_fail_loudly(gpu):
    from vcvits_amd import mel_processing
    y = torch.zeros(1, 4000, device=gpu)
    with pytest.raises(NotImplementedError):
        mel_processing.spectrogram_torch(y, 1001, 22050, 250, 1001)  # odd
    with pytest.raises(NotImplementedError):
        mel_processing.spectrogram_torch(y, 8192, 22050, 2048, 8192)  # beyond 4096
    with pytest.raises(ValueError):
        mel_processing.spectrogram_torch(y, 1024, 22050, 256, 2048)  # win_length > n_fft (torch.stft refuses it too)


@pytest.mark.gpu
def test_pipeline_at_other_sizes(gpu):
    """vits/model/pipeline.py:11-70 (its own constructor defaults are n_fft 1024 / win 1024 / hop 256): complex STFT ->
    inverse STFT at the tuned size with a shorter window, and at other sizes on the generic kernels."""
    from vcvits_amd import ops
    from vcvits_amd.model.pipeline import SpeechConversionAudioPipeline
    rng = np.random.default_rng(9)
    wav = torch.from_numpy((rng.standard_normal((2, 1, 16000)) * 0.2).astype(np.float32))
    for n_fft, hop, win in ((2048, 512, 2048), (2048, 512, 1600), (2048, 512, 1024), (1024, 256, 1024), (512, 128, 400),
                            (4096, 1024, 4096), (256, 64, 256), (1280, 320, 1280), (400, 100, 400), (3000, 750, 2400)):
        pipe = SpeechConversionAudioPipeline(sr=16000, n_fft=n_fft, n_mel=128, win_length=win, hop_length=hop)
        out = pipe(wav.to(gpu))
        ref = O.audio_pipeline(wav, n_fft=n_fft, hop_length=hop, win_length=win)
        close("pipeline %d/%d/%d" % (n_fft, hop, win), out, ref, tol=1e-5, atol=2e-6)
        # the complex spectrum itself (torch.stft on the zero-padded signal)
        pad = (n_fft - hop) // 2
        spec = ops.stft_complex(wav[:, 0].to(gpu), n_fft, hop, pad, reflect=False, win_length=win)
        yp = torch.nn.functional.pad(wav[:, 0], (pad, pad))
        sref = torch.stft(yp, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=False,
                          return_complex=True)
        close("stft %d/%d/%d" % (n_fft, hop, win), torch.view_as_real(spec), torch.view_as_real(sref), tol=1e-5)
    default = SpeechConversionAudioPipeline()  # the reference's own defaults construct and run
    assert default(wav.to(gpu)).shape == wav.shape
