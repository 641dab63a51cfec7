"""vits/model/encoders/posterior_encoder.py:9-39"""
import torch
from torch import nn

from ... import commons, ops
from .. import modules


class PosteriorEncoder(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                 gin_channels=0):
        super().__init__()
        self.in_channels, self.out_channels, self.hidden_channels = in_channels, out_channels, hidden_channels
        self.kernel_size, self.dilation_rate, self.n_layers, self.gin_channels = kernel_size, dilation_rate, n_layers, gin_channels
        self.pre = modules.Conv(in_channels, hidden_channels, 1)
        self.enc = modules.WN(hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=gin_channels)
        self.proj = modules.Conv(hidden_channels, out_channels * 2, 1)

    def forward(self, x, x_lengths, g=None, noise=None):
        """`noise` optionally injects the torch.randn_like draw (parity tests)."""
        x_mask = torch.unsqueeze(commons.sequence_mask(x_lengths, x.size(2)), 1).to(x.dtype)
        mask2 = x_mask.reshape(x_mask.shape[0], -1)
        h = ops.mask_mul(self.pre(x), mask2)
        h = self.enc(h, x_mask, g=g)
        stats = self.proj(h)
        if noise is None:
            noise = torch.randn((x.shape[0], self.out_channels, x.shape[2]), device=x.device, dtype=x.dtype)
        z, m, logs = ops.posterior_sample(stats, noise, mask2)
        return z, m, logs, x_mask
