"""Content (prior) encoder (vits/model/encoders/content_encoder.py:14-73 and :76-126).

The frozen fairseq HuBERT (:32-35) is third-party and absent offline; its CONTRACT (:53-56) is built: a
pluggable `feature_extractor` receives the source waveform padded by (400 - 320) // 2 samples per side and
returns frame features [B, T', hubert_channels] (or the fairseq `(features, padding_mask)` tuple), which are
transposed to [B, hubert_channels, T'] and fed to the HIP path.  Without an extractor the forward takes
precomputed / synthetic features [B, hubert_channels, T] directly (every benchmark configuration does)."""
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


HUBERT_WINDOW, HUBERT_DOWNSAMPLE = 400, 320  # receptive field / hop of HuBERT's conv front end (content_encoder.py:54)


class HubertContentEncoder(nn.Module):
    """Same parameters/keys as the reference class minus the frozen ``hubert.*`` buffers.
    ``forward(x, x_lengths, pitch, pitch_lengths)``: x is the source waveform [B, 1, T] (reference signature;
    needs a feature extractor) or HuBERT features [B, hubert_channels, T']."""
    concat = False

    def __init__(self, hubert_ckpt, out_channels, hidden_channels, filter_channels, n_heads, n_layers,
                 kernel_size, p_dropout, hubert_channels, num_pitch, n_fft=2048, hop_size=512, feature_extractor=None):
        super().__init__()
        self.n_fft, self.hop_size, self.out_channels = n_fft, hop_size, out_channels
        self.hubert_channels = hubert_channels
        # kept out of the module tree (object.__setattr__): a frozen third-party model must not add `hubert.*`
        # keys to this module's state_dict or parameters to the optimizer
        object.__setattr__(self, "_extractor", feature_extractor)
        proj_channels = hidden_channels // 2 if self.concat else hidden_channels
        self.hubert_proj = _Linear(hubert_channels, proj_channels)
        self.emb_pitch = nn.Embedding(num_pitch, proj_channels)
        nn.init.normal_(self.emb_pitch.weight, 0.0, proj_channels ** -0.5)
        if self.concat:
            self.pitch_proj = _Linear(proj_channels, proj_channels)
        self.encoder = TransformerEncoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout)
        self.proj = Conv(hidden_channels, out_channels * 2, 1)

    def set_feature_extractor(self, extractor):
        """extractor: object with `.extract_features(wav [B, T]) -> (feats [B, T', H], padding_mask)` (the fairseq
        HubertModel contract the reference calls, content_encoder.py:55) or a plain callable wav -> feats."""
        object.__setattr__(self, "_extractor", extractor)

    def extract(self, wav):
        """content_encoder.py:53-56: pad 40 + 40 samples, frozen feature extractor, [B, T', H] -> [B, H, T']."""
        ex = self._extractor
        if ex is None:
            raise RuntimeError("HubertContentEncoder: a waveform batch [B, 1, T] needs a feature extractor "
                               "(set_feature_extractor); without one pass HuBERT features [B, %d, T']"
                               % self.hubert_channels)
        pad = (HUBERT_WINDOW - HUBERT_DOWNSAMPLE) // 2
        with torch.no_grad():
            w = F.pad(wav, (pad, pad)).squeeze(1)
            out = ex.extract_features(w) if hasattr(ex, "extract_features") else ex(w)
            feats = out[0] if isinstance(out, (tuple, list)) else out
            if feats.dim() != 3 or feats.shape[2] != self.hubert_channels:
                raise RuntimeError("feature extractor returned %s, expected [B, T', %d]"
                                   % (tuple(feats.shape), self.hubert_channels))
            return feats.transpose(1, -1).float().contiguous()

    def forward(self, x, x_lengths, pitch, pitch_lengths):
        if x.dim() == 3 and x.shape[1] == 1 and self.hubert_channels != 1:
            x = self.extract(x)
        hubert_out = self.hubert_proj(x)
        pitch_out = ops.embedding_t(pitch, self.emb_pitch.weight)  # [B, proj, T]: emb_pitch(pitch).transpose(1, -1)
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

    def forward(self, x, x_lengths, pitch, pitch_lengths):
        if x.dim() == 3 and x.shape[1] == 1 and self.hubert_channels != 1:
            raise RuntimeError("PreloadHubertContentEncoder takes precomputed features (content_encoder.py:110)")
        return super().forward(x, x_lengths, pitch, pitch_lengths)
