"""vits/model/synthesizers/synthesizer_svc.py:18-119 of the reference.

Differences forced by what is outside the tree: the frozen fairseq HuBERT (:57, content_encoder.py:54-56) is a
pluggable `feature_extractor` (kwarg, or `enc_p.set_feature_extractor`); without one the content encoder takes
HuBERT *features* [B, hubert_channels, T] instead of a waveform.  The decoder is the in-package HiFi-GAN
Generator built from the config's own hyper-parameters where the reference downloads one from torch.hub (:59)."""
import torch
from torch import nn

from ... import commons, ops
from ..encoders.content_encoder import HubertContentEncoder, PreloadHubertContentEncoder
from ..encoders.posterior_encoder import PosteriorEncoder
from ..flow import ResidualCouplingBlock
from ..generator import Generator


class SynthesizerSVC(nn.Module):
    def __init__(self, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, hubert_channels, num_pitch,
                 n_speakers=0, gin_channels=0, **kwargs):
        super().__init__()
        self.spec_channels, self.inter_channels, self.hidden_channels = spec_channels, inter_channels, hidden_channels
        self.filter_channels, self.n_heads, self.n_layers, self.kernel_size = filter_channels, n_heads, n_layers, kernel_size
        self.p_dropout, self.resblock = p_dropout, resblock
        self.resblock_kernel_sizes, self.resblock_dilation_sizes = resblock_kernel_sizes, resblock_dilation_sizes
        self.upsample_rates, self.upsample_initial_channel = upsample_rates, upsample_initial_channel
        self.upsample_kernel_sizes, self.segment_size = upsample_kernel_sizes, segment_size
        self.n_speakers, self.gin_channels = n_speakers, gin_channels
        self.hubert_channels, self.num_pitch = hubert_channels, num_pitch

        if kwargs.get("content_encoder", "hubert") == "preload":
            self.enc_p = PreloadHubertContentEncoder(inter_channels, hidden_channels, filter_channels, n_heads,
                                                     n_layers, kernel_size, p_dropout, hubert_channels, num_pitch)
        else:
            self.enc_p = HubertContentEncoder(kwargs.get("hubert_ckpt"), inter_channels, hidden_channels,
                                              filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                                              hubert_channels, num_pitch,
                                              feature_extractor=kwargs.get("feature_extractor"))
        self.dec = Generator(inter_channels, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                             upsample_rates, upsample_initial_channel, upsample_kernel_sizes,
                             gin_channels=kwargs.get("dec_gin_channels", 0))
        self.enc_q = PosteriorEncoder(spec_channels, inter_channels, hidden_channels, 5, 1, 16, gin_channels=gin_channels)
        self.flow = ResidualCouplingBlock(inter_channels, hidden_channels, 5, 1, 4, gin_channels=gin_channels)
        if n_speakers >= 1:
            self.emb_g = nn.Embedding(n_speakers, gin_channels)

    def _g(self, sid):
        if self.n_speakers >= 1:
            return ops.embedding_t(sid, self.emb_g.weight)  # [b, h, 1] = emb_g(sid).unsqueeze(-1)
        return None

    def forward(self, x_wav, x_wav_lengths, x_pitch, x_pitch_lengths, y_spec, y_spec_lengths, sid=None,
                noise=None, ids_slice=None, decoder_only=False, raw_sizes=None):
        """`noise` / `ids_slice` optionally inject the two random draws (parity tests).  decoder_only: compute only what
        `o` depends on -- posterior encoder, slice, decoder -- and return None for the prior-side results: the reference's
        discriminator step runs this whole forward under no_grad and keeps `y_hat.detach()` alone (vcvits.py:119,153), so
        its content encoder and flow are dead work there.  raw_sizes: see data/collate.py: bucket_batch."""
        g = self._g(sid)
        if decoder_only:
            x_mask = m_p = logs_p = z_p = None
            z, m_q, logs_q, y_mask = self.enc_q(y_spec, y_spec_lengths, g=g, noise=noise)
        else:
            x, m_p, logs_p, x_mask = self.enc_p(x_wav, x_wav_lengths, x_pitch, x_pitch_lengths)
            z, m_q, logs_q, y_mask = self.enc_q(y_spec, y_spec_lengths, g=g, noise=noise)
            z_p = self.flow(z, y_mask, g=g)
            # (raw_sizes: a length-bucketed batch keeps the alignment of its own un-bucketed padding, data/collate.py)
            m_p = ops.interpolate_nearest(m_p, y_spec.shape[2], raw_sizes)
            logs_p = ops.interpolate_nearest(logs_p, y_spec.shape[2], raw_sizes)
        if ids_slice is None:
            z_slice, ids_slice = commons.rand_slice_segments(z, y_spec_lengths, self.segment_size)
        else:
            z_slice = commons.slice_segments(z, ids_slice, self.segment_size)
        o = self.dec(z_slice)
        return o, ids_slice, z_slice, x_mask, y_mask, (z, z_p, m_p, logs_p, m_q, logs_q)

    def infer(self, x, x_lengths, x_pitch, x_pitch_lengths, sid=None, noise_scale=1, length_scale=1,
              noise_scale_w=1., max_len=None, noise=None):
        x, m_p, logs_p, x_mask = self.enc_p(x, x_lengths, x_pitch, x_pitch_lengths)
        g = self._g(sid)
        y_lengths = (x_lengths * length_scale).long()
        y_mask = torch.unsqueeze(commons.sequence_mask(y_lengths, None), 1).to(x_mask.dtype)
        y_max_len = int(torch.max(y_lengths).item())
        m_p = ops.interpolate_nearest(m_p, y_max_len)
        logs_p = ops.interpolate_nearest(logs_p, y_max_len)
        if noise is None:
            noise = torch.randn_like(m_p)
        z_p = ops.prior_sample(m_p, logs_p, noise, float(noise_scale))  # m_p + noise * exp(logs_p) * noise_scale
        z = self.flow(z_p, y_mask, g=g, reverse=True)
        zm = ops.mask_mul(z, y_mask.reshape(y_mask.shape[0], -1))
        o = self.dec(zm[:, :, :max_len].contiguous())
        return o, y_mask, (z, z_p, m_p, logs_p)

    def voice_conversion(self, y, y_lengths, sid_src, sid_tgt):
        assert self.n_speakers > 0, "n_speakers have to be larger than 0."
        g_src = ops.embedding_t(sid_src, self.emb_g.weight)
        g_tgt = ops.embedding_t(sid_tgt, self.emb_g.weight)
        z, m_q, logs_q, y_mask = self.enc_q(y, y_lengths, g=g_src)
        z_p = self.flow(z, y_mask, g=g_src)
        z_hat = self.flow(z_p, y_mask, g=g_tgt, reverse=True)
        zm = ops.mask_mul(z_hat, y_mask.reshape(y_mask.shape[0], -1))
        o_hat = self.dec(zm, g=g_tgt if hasattr(self.dec, "cond") else None)
        return o_hat, y_mask, (z, z_p, z_hat)


SynthesizerTrn = SynthesizerSVC  # the name BASELINE.json's north_star uses
