"""Content (prior) encoder, post-HuBERT part (vits/model/encoders/content_encoder.py:58-73 and
:110-126).  HuBERT itself (fairseq, frozen, :32-35,:54-56) is out of scope: every benchmark
configuration feeds precomputed / synthetic features [B, hubert_channels, T]."""
import torch
from torch import nn
from torch.nn import functional as F

from ... import commons, ops
from ..modules import Conv
from ..transformer.relative_attention_transformer import TransformerEncoder


class _Linear(nn.Module):
    """nn.Linear parameters ([out, in] weight), applied to [B, in, T] as a 1x1 conv on the MFMA
    kernel (same arithmetic as F.linear on the transposed tensor)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        ref = nn.Linear(in_features, out_features)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())

    def forward(self, x):
        return ops.conv1d(x, self.weight.unsqueeze(-1), self.bias)


class HubertContentEncoder(nn.Module):
    """Same parameters/keys as the reference class minus the frozen ``hubert.*`` buffers; forward
    takes the HuBERT features directly: ``forward(feats [B,hubert,T], x_lengths, pitch, pitch_lengths)``."""
    concat = False

    def __init__(self, hubert_ckpt, out_channels, hidden_channels, filter_channels, n_heads, n_layers,
                 kernel_size, p_dropout, hubert_channels, num_pitch, n_fft=2048, hop_size=512):
        super().__init__()
        self.n_fft, self.hop_size, self.out_channels = n_fft, hop_size, out_channels
        proj_channels = hidden_channels // 2 if self.concat else hidden_channels
        self.hubert_proj = _Linear(hubert_channels, proj_channels)
        self.emb_pitch = nn.Embedding(num_pitch, proj_channels)
        nn.init.normal_(self.emb_pitch.weight, 0.0, proj_channels ** -0.5)
        if self.concat:
            self.pitch_proj = _Linear(proj_channels, proj_channels)
        self.encoder = TransformerEncoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.proj = Conv(hidden_channels, out_channels * 2, 1)

    def forward(self, x, x_lengths, pitch, pitch_lengths):
        hubert_out = self.hubert_proj(x)
        pitch_out = F.embedding(pitch, self.emb_pitch.weight).transpose(1, -1).contiguous()  # gather: torch glue
        if self.concat:
            out = torch.cat((hubert_out, self.pitch_proj(pitch_out)), dim=1)
        else:
            out = hubert_out + pitch_out
        # quirk kept: x_lengths are whatever the caller passes (sample counts in the reference's
        # training loop, so the mask is all ones there -- SURVEY.md Appendix D.8)
        x_mask = torch.unsqueeze(commons.sequence_mask(x_lengths.int(), out.size(2)), 1).to(x.dtype)
        mask2 = x_mask.reshape(x_mask.shape[0], -1)
        x_out = self.encoder(ops.mask_mul(out, mask2), x_mask)
        m, logs = ops.split_stats(self.proj(x_out), mask2)
        return x_out, m, logs, x_mask


class PreloadHubertContentEncoder(HubertContentEncoder):
    """content_encoder.py:76-126: half-width projections concatenated."""
    concat = True

    def __init__(self, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                 hubert_channels, num_pitch, n_fft=2048, hop_size=512):
        super().__init__(None, out_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size,
                         p_dropout, hubert_channels, num_pitch, n_fft, hop_size)
