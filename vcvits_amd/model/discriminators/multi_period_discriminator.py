"""vits/model/discriminators/multi_period_discriminator.py:9-31 (one DiscriminatorS + one
DiscriminatorP per period)."""
from typing import List

from torch import nn

from ..modules import prepare_weight_norm
from ._pair import run_many, streams
from .discriminator import DiscriminatorP, DiscriminatorS


class MultiPeriodDiscriminator(nn.Module):
    def __init__(self, periods: List[int] = [2, 3, 5, 7, 11, 17, 23, 37], use_spectral_norm: bool = False):
        super().__init__()
        self.periods = periods
        discs = [DiscriminatorS(use_spectral_norm=use_spectral_norm)]
        discs = discs + [DiscriminatorP(i, use_spectral_norm=use_spectral_norm) for i in periods]
        self.discriminators = nn.ModuleList(discs)

    def forward(self, y, y_hat, g=None):
        if streams() <= 1:
            prepare_weight_norm(self)  # one launch for every layer of every sub-discriminator
            for d in self.discriminators:
                d._wn_parent_prepared = True
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        for y_d_r, y_d_g, fmap_r, fmap_g in run_many(self.discriminators, [(y, y_hat)] * len(self.discriminators)):
            y_d_rs.append(y_d_r)
            y_d_gs.append(y_d_g)
            fmap_rs.append(fmap_r)
            fmap_gs.append(fmap_g)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs
