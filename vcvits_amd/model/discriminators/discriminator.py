"""vits/model/discriminators/discriminator.py: DiscriminatorP :12-46, DiscriminatorS :49-74.
Every conv (+ its leaky-ReLU) is one launch of the MFMA conv kernel; the (k,1) Conv2d of the
period discriminator is addressed as rows of `period` columns, the reflect pad is one kernel."""
import torch
from torch import nn

from ... import ops
from ..._lib import ACT_LEAKY
from ...commons import get_padding
from ..modules import Conv, LRELU_SLOPE, prepare_weight_norm


class DiscriminatorP(nn.Module):
    def __init__(self, period, kernel_size=5, stride=3, use_spectral_norm=False):
        super().__init__()
        self.use_spectral_norm = sn = bool(use_spectral_norm)  # norm_f = spectral_norm instead of weight_norm (:17)
        self.period = period
        pad = get_padding(kernel_size, 1)
        chans = [1, 32, 128, 512, 1024, 1024]
        self.convs = nn.ModuleList([
            Conv(chans[i], chans[i + 1], kernel_size, stride=(stride if i < 4 else 1), padding=pad, weight_norm=True,
                 two_d=True, spectral_norm=sn) for i in range(5)])
        self.conv_post = Conv(1024, 1, 3, padding=1, weight_norm=True, two_d=True, spectral_norm=sn)

    def forward(self, x):
        prepare_weight_norm(self)
        fmap = []
        b, c, t = x.shape
        if t % self.period != 0:
            n_pad = self.period - (t % self.period)
            x = ops.reflect_pad_right(x, n_pad)
            t = t + n_pad
        x = x.view(b, c, t // self.period, self.period)
        for l in self.convs:
            x, rec = ops.fmap_tap(l(x, out_act=ACT_LEAKY, slope=LRELU_SLOPE))
            fmap.append(rec)
        x, rec = ops.fmap_tap(self.conv_post(x))
        fmap.append(rec)
        return torch.flatten(x, 1, -1), fmap


class DiscriminatorS(nn.Module):
    def __init__(self, use_spectral_norm=False):
        super().__init__()
        self.use_spectral_norm = sn = bool(use_spectral_norm)  # (:52)
        cfg = [(1, 16, 15, 1, 7, 1), (16, 64, 41, 4, 20, 4), (64, 256, 41, 4, 20, 16), (256, 1024, 41, 4, 20, 64),
               (1024, 1024, 41, 4, 20, 256), (1024, 1024, 5, 1, 2, 1)]
        self.convs = nn.ModuleList([Conv(ci, co, k, stride=s, padding=p, groups=g, weight_norm=True, spectral_norm=sn)
                                    for ci, co, k, s, p, g in cfg])
        self.conv_post = Conv(1024, 1, 3, padding=1, weight_norm=True, spectral_norm=sn)

    def forward(self, x):
        prepare_weight_norm(self)
        fmap = []
        for l in self.convs:
            x, rec = ops.fmap_tap(l(x, out_act=ACT_LEAKY, slope=LRELU_SLOPE))
            fmap.append(rec)
        x, rec = ops.fmap_tap(self.conv_post(x))
        fmap.append(rec)
        return torch.flatten(x, 1, -1), fmap
