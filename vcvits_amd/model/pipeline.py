"""Source-audio front end of the reference (vits/model/pipeline.py:11-70): complex spectrogram ->
optional SpecAugment frequency mask -> inverse spectrogram, copied into a zero tensor of the input's
shape.  torchaudio's Spectrogram / InverseSpectrogram are restated on the HIP STFT / iSTFT kernels
(torchaudio itself is absent from the reference tree and from this image: parity is checked against
torch.stft / torch.istft).  Runs without autograd, as the reference does (vcvits.py:61-62)."""
import random

import torch
from torch import nn

from .. import ops


class SpeechConversionAudioPipeline(nn.Module):
    def __init__(self, sr=16000, n_fft=1024, n_mel=128, win_length=1024, hop_length=256):
        super().__init__()
        # (n_fft = 2048, both reference configs: the tuned kernels; any other power of two in [64, 4096]: the generic ones;
        # win_length <= n_fft either way.  Anything else raises NotImplementedError from ops.stft_complex when called)
        self.source_sampling_rate = sr
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.pad = int((n_fft - hop_length) / 2)
        self.freq_mask_param = 80

    def forward(self, waveform: torch.Tensor, aug: bool = False) -> torch.Tensor:
        b, c, t = waveform.shape
        spec = ops.stft_complex(waveform.reshape(b * c, t), self.n_fft, self.hop_length, self.pad, reflect=False,
                                win_length=self.win_length)
        if aug:
            # torchaudio.transforms.FrequencyMasking(80): one random band [f0, f0+f) zeroed for the whole batch
            f = int(random.random() * self.freq_mask_param)
            f0 = int(random.random() * (spec.shape[1] - f))
            spec = spec.clone()
            spec[:, f0:f0 + f] = 0
        wav = ops.istft(spec, self.n_fft, self.hop_length, center=True, win_length=self.win_length).reshape(b, c, -1)
        out = torch.zeros_like(waveform)
        n = min(wav.shape[2], t)
        out[:, :, :n] = wav[:, :, :n]
        return out
