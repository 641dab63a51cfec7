"""Pitch binning and inference plumbing of the data path (reference: vits/data/audio.py:65-76, infer.py:81).

Decoding audio files and pYIN pitch tracking stay with the reference's third-party stack (torchaudio,
librosa: not installed here); the dataset below reads what that stack cached."""
import math

import torch


def coarse_f0(f0, f0_min=50.0, f0_max=1100.0, f0_bin=512):
    """Hz -> integer pitch class in [1, f0_bin-1] on a mel scale (vits/data/audio.py:65-76): voiced frames map
    linearly in mel between f0_min and f0_max onto [1, f0_bin-1]; unvoiced (f0 = 0 -> mel 0) and anything that
    lands at or below 1 become class 1; values past the top class saturate; half-to-even rounding (torch.round).
    Returns a float tensor of whole numbers, as the reference does (callers cast with .long())."""
    f0 = torch.as_tensor(f0, dtype=torch.float32)
    mel_min = 1127.0 * math.log(1.0 + f0_min / 700.0)
    mel_max = 1127.0 * math.log(1.0 + f0_max / 700.0)
    mel = 1127.0 * torch.log(1.0 + f0 / 700.0)
    scaled = (mel - mel_min) * (f0_bin - 2) / (mel_max - mel_min) + 1.0
    mel = torch.where(mel > 0, scaled, mel)
    mel = torch.clamp(mel, min=1.0, max=float(f0_bin - 1))
    out = torch.round(mel)
    if out.numel() and not (float(out.max()) < f0_bin and float(out.min()) >= 1):
        raise AssertionError((float(out.max()), float(out.min())))
    return out


def infer_length_scale(data_hparams):
    """Frames of the target rate per source sample (infer.py:81): the `length_scale` handed to
    SynthesizerSVC.infer so the content features (source rate / 320) are resampled to target frames."""
    return (data_hparams.target_sampling_rate / data_hparams.hop_length) / data_hparams.source_sampling_rate
