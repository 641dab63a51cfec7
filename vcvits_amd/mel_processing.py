"""vits/mel_processing.py of the reference on the HIP STFT / mel kernels.

spectrogram_torch (reflect pad, :54-74), spectrogram_torch_audio (torchaudio spectrogram =
zero pad, :76-96), spec_to_mel_torch (:98-112), mel_spectrogram_torch (:115-142).  The mel
filterbank restates librosa.filters.mel(htk=False, norm='slaney')."""
import logging
import os

import numpy as np
import torch

from . import ops, tuning

MAX_WAV_VALUE = 32768.0
_CHECK_RANGE = tuning.flag("VCVITS_CHECK_RANGE", False, "mel_processing: log when a waveform leaves [-1, 1] (the reference prints; a device sync)")

mel_basis = {}


def _range_warn(y):
    # the reference logs a warning after two host syncs per call (:55-58); opt-in here
    if _CHECK_RANGE:
        lo, hi = float(y.min()), float(y.max())
        if lo < -1.:
            logging.warning(f'min value is {lo}')
        if hi > 1.:
            logging.warning(f'max value is {hi}')


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * 3.0 / 200.0
    log_region = f >= 1000.0
    out = lin.copy()
    out[log_region] = 15.0 + np.log(f[log_region] / 1000.0) * (27.0 / np.log(6.4))
    return out


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    out = m * 200.0 / 3.0
    log_region = m >= 15.0
    out[log_region] = 1000.0 * np.exp((m[log_region] - 15.0) * (np.log(6.4) / 27.0))
    return out


def librosa_mel_fn(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """Slaney-normalised triangular filterbank [n_mels, n_fft/2+1], float32."""
    if fmax is None:
        fmax = sr / 2.0
    freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(np.array([fmin]))[0],
                                          _slaney_hz_to_mel(np.array([fmax]))[0], n_mels + 2))
    width = np.diff(edges)
    ramp = edges[:, None] - freqs[None, :]
    rising = -ramp[:-2] / width[:-1, None]
    falling = ramp[2:] / width[1:, None]
    fb = np.maximum(0.0, np.minimum(rising, falling))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb.astype(np.float32)


def _mel_matrix(n_fft, num_mels, sampling_rate, fmin, fmax, dtype, device):
    key = "%s_%s_%s_%s_%s_%s_%s" % (fmax, fmin, num_mels, n_fft, sampling_rate, dtype, device)
    if key not in mel_basis:
        mel = librosa_mel_fn(sr=sampling_rate, n_fft=n_fft, n_mels=num_mels, fmin=fmin, fmax=fmax)
        mel_basis[key] = torch.from_numpy(mel).to(dtype=dtype, device=device).contiguous()
    return mel_basis[key]


def _spec(y, n_fft, hop_size, win_size, reflect):
    _range_warn(y)
    lead = y.shape[:-1]
    y2 = y.reshape(-1, y.shape[-1])
    mag = ops.stft_mag(y2, n_fft, hop_size, int((n_fft - hop_size) / 2), reflect, 1e-6, win_length=win_size)
    return mag.reshape(lead + mag.shape[-2:])


def spectrogram_torch(y, n_fft: int, sampling_rate: int, hop_size: int, win_size: int, center: bool = False):
    return _spec(y, n_fft, hop_size, win_size, True)


def spectrogram_torch_audio(y, n_fft: int, sampling_rate: int, hop_size: int, win_size: int, center: bool = False):
    return _spec(y, n_fft, hop_size, win_size, False)


def spec_to_mel_torch(spec, n_fft, num_mels, sampling_rate, fmin, fmax):
    mel = _mel_matrix(n_fft, num_mels, sampling_rate, fmin, fmax, spec.dtype, spec.device)
    lead = spec.shape[:-2]
    s3 = spec.reshape((-1,) + spec.shape[-2:])
    out = ops.mel_log(s3, mel, 1e-5)
    return out.reshape(lead + out.shape[-2:])


def mel_spectrogram_torch(y, n_fft: int, num_mels: int, sampling_rate: int, hop_size: int, win_size: int,
                          fmin: int, fmax: int, center: bool = False):
    spec = spectrogram_torch(y, n_fft, sampling_rate, hop_size, win_size, center)
    return spec_to_mel_torch(spec, n_fft, num_mels, sampling_rate, fmin, fmax)
