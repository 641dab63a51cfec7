"""vits/model/discriminators/multi_scale_discriminator.py:10-42 (5 DiscriminatorS on inputs
pooled by cascaded AvgPool1d(4, 2, 2))."""
from torch import nn

from ... import ops
from ..modules import prepare_weight_norm
from ._pair import run_many, streams
from .discriminator import DiscriminatorS


class MultiScaleDiscriminator(nn.Module):
    def __init__(self, use_spectral_norm=False):
        super().__init__()
        self.discriminators = nn.ModuleList(
            [DiscriminatorS(use_spectral_norm=use_spectral_norm)] + [DiscriminatorS() for _ in range(4)])

    def forward(self, y, y_hat):
        if streams() <= 1:
            prepare_weight_norm(self)  # one launch for every layer of every sub-discriminator
            for d in self.discriminators:
                d._wn_parent_prepared = True
        y_d_rs, y_d_gs, fmap_rs, fmap_gs = [], [], [], []
        inputs = []
        for i in range(len(self.discriminators)):
            if i != 0:
                y = ops.avgpool4(y)
                y_hat = ops.avgpool4(y_hat)
            inputs.append((y, y_hat))
        for y_d_r, y_d_g, fmap_r, fmap_g in run_many(self.discriminators, inputs):
            y_d_rs.append(y_d_r)
            fmap_rs.append(fmap_r)
            y_d_gs.append(y_d_g)
            fmap_gs.append(fmap_g)
        return y_d_rs, y_d_gs, fmap_rs, fmap_gs
