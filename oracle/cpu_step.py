"""CPU oracle of one training batch (generator step, then discriminator step) -- TEST
INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).

Restates vits/light/vcvits.py:54-183 + :247-257 of the reference with the functional oracle
(oracle/vits_oracle.py), torch-CPU autograd and torch.optim.AdamW, on the same state_dict keys the
product modules use.  `vocoder_gan_batch` is the BASELINE.json configs[1] workload (decoder +
MPD + MSD + STFT/mel L1); `full_batch` is the whole SynthesizerSVC step.
"""
import torch

from . import vits_oracle as O


def _is_buffer(sd, k):
    """The power-iteration vectors of a spectrally normed layer (buffers of torch.nn.utils.spectral_norm, not parameters):
    weight_u, and weight_v where the layer carries weight_orig (weight_v of a weight-normed layer IS a parameter)."""
    return k.endswith(".weight_u") or (k.endswith(".weight_v") and k[:-1] + "orig" in sd)


def _leaf_copy(sd, dtype=torch.float32):
    return {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone())
            .requires_grad_(v.is_floating_point() and not _is_buffer(sd, k)) for k, v in sd.items()}


class CpuTrainer:
    def __init__(self, module_state_dict, hparams, periods, vocoder_only, dtype=torch.float32):
        """module_state_dict: state_dict of the product VCVITS / VocoderGAN module (CPU tensors).  dtype=torch.float64: the
        same step in double precision -- the yardstick the fp64 ranking tests measure both fp32 implementations against
        (tests/test_full_width_step_gpu.py)."""
        self.hp = hparams
        self.periods = list(periods)
        self.vocoder_only = vocoder_only
        self.dtype = dtype
        self.sd = _leaf_copy(module_state_dict, dtype)
        g_keys = [k for k in self.sd if k.startswith("net_g.") and self.sd[k].requires_grad]
        d_keys = [k for k in self.sd if (k.startswith("net_period_d.") or k.startswith("net_scale_d."))
                  and self.sd[k].requires_grad]
        t = hparams["train"]
        self.g_params = [self.sd[k] for k in g_keys]
        self.d_params = [self.sd[k] for k in d_keys]
        self.opt_g = torch.optim.AdamW(self.g_params, t["learning_rate"], betas=tuple(t["betas"]), eps=t["eps"])
        self.opt_d = torch.optim.AdamW(self.d_params, t["learning_rate"], betas=tuple(t["betas"]), eps=t["eps"])
        d = hparams["data"]
        self.melmat = torch.from_numpy(O.mel_filterbank(d["target_sampling_rate"], d["filter_length"],
                                                        d["n_mel_channels"], d["mel_fmin"], d["mel_fmax"])).to(dtype)
        # dropout of the content encoder: None = identity; tests set a callable that multiplies by the masks the HIP
        # step drew (same elements dropped on both sides), consumed in the reference's call order
        self.drop = None

    def _mel(self, wav):
        d = self.hp["data"]
        spec = O.spectrogram(wav, d["filter_length"], d["hop_length"], d["win_length"], reflect=False)
        return spec, O.spec_to_mel(spec, self.melmat)

    def _dec(self, z):
        m = self.hp["model"]
        prefix = "net_g" if self.vocoder_only else "net_g.dec"
        return O.generator_forward(self.sd, prefix, z, m["upsample_rates"], m["upsample_kernel_sizes"],
                                   m["resblock_kernel_sizes"], m["resblock_dilation_sizes"], resblock=m.get("resblock", "1"))

    def _generator_pass(self, batch):
        d, t, m = self.hp["data"], self.hp["train"], self.hp["model"]
        if self.vocoder_only:
            y = batch["y_wav_values"]
            y_hat = self._dec(batch["z_slice"])
            with torch.no_grad():
                _, y_mel = self._mel(y.squeeze(1))
            return y_hat, y, y_mel, None
        sd = self.sd
        C, H = m["inter_channels"], m["hidden_channels"]
        with torch.no_grad():
            y_spec, y_mel = self._mel(batch["y_wav_values"].squeeze(1))
            y_spec_lengths = (batch["y_wav_lengths"] / d["hop_length"]).long()
        x, m_p, logs_p, x_mask = O.content_encoder_forward(
            sd, "net_g.enc_p", batch["x_hubert_features_values"], batch["x_hubert_features_lengths"],
            batch["x_pitch_values"], C, m["n_heads"], m["n_layers"], m["kernel_size"], drop=self.drop)
        g = torch.nn.functional.embedding(batch["sid"], sd["net_g.emb_g.weight"]).unsqueeze(-1)
        z, m_q, logs_q, y_mask = O.posterior_encoder_forward(sd, "net_g.enc_q", y_spec, y_spec_lengths, g,
                                                            batch["noise"], C, H, 5, 1, 16)
        z_p = O.flow_forward(sd, "net_g.flow", z, y_mask, g, False, C, H, 5, 1, 4)
        T_y = y_spec.shape[2]
        m_p = torch.nn.functional.interpolate(m_p, size=(T_y,), mode="nearest")
        logs_p = torch.nn.functional.interpolate(logs_p, size=(T_y,), mode="nearest")
        ids = batch["ids_slice"]
        seg = t["segment_size"] // d["hop_length"]
        z_slice = O.slice_segments(z, ids, seg)
        y_hat = self._dec(z_slice)
        with torch.no_grad():
            y = O.slice_segments(batch["y_wav_values"], ids * d["hop_length"], t["segment_size"])
            y_mel_slice = O.slice_segments(y_mel, ids, seg)
        return y_hat, y, y_mel_slice, (z_p, logs_q, m_p, logs_p, y_mask)

    def batch(self, batch_g, batch_d=None):
        """One reference batch: G step on batch_g then D step on batch_d (same data; the random
        draws may differ between the two passes).  Returns the two losses."""
        t = self.hp["train"]
        cast = lambda b: {k: (v.to(self.dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in b.items()}
        batch_g = cast(batch_g)
        batch_d = cast(batch_d) if batch_d is not None else batch_g
        for p in self.d_params:
            p.requires_grad_(False)
        for p in self.g_params:
            p.requires_grad_(True)
        self.opt_g.zero_grad()
        y_hat, y, y_mel, kl_args = self._generator_pass(batch_g)
        _, gp, frp, fgp = O.mpd_forward(self.sd, "net_period_d", y, y_hat, self.periods, training=True)
        _, gs, frs, fgs = O.msd_forward(self.sd, "net_scale_d", y, y_hat, training=True)
        _, mel_hat = self._mel(y_hat.squeeze(1))
        loss_g = (O.generator_loss(gs) + O.feature_loss(frs, fgs)) + (O.generator_loss(gp) + O.feature_loss(frp, fgp)) \
            + torch.nn.functional.l1_loss(mel_hat, y_mel) * t["c_mel"]
        if kl_args is not None:
            loss_g = loss_g + O.kl_loss(*kl_args) * t["c_kl"]
        loss_g.backward()
        self.grads_g = {k: self.sd[k].grad.detach().clone() for k in self.sd
                        if self.sd[k].requires_grad and self.sd[k].grad is not None}
        self.opt_g.step()
        for p in self.d_params:
            p.requires_grad_(True)
        for p in self.g_params:
            p.requires_grad_(False)
        self.opt_d.zero_grad()
        with torch.no_grad():
            y_hat, y, _, _ = self._generator_pass(batch_d)
        rp, gp, _, _ = O.mpd_forward(self.sd, "net_period_d", y, y_hat.detach(), self.periods, training=True)
        rs, gs, _, _ = O.msd_forward(self.sd, "net_scale_d", y, y_hat.detach(), training=True)
        loss_d = O.discriminator_loss(rp, gp) + O.discriminator_loss(rs, gs)
        loss_d.backward()
        self.grads_d = {k: self.sd[k].grad.detach().clone() for k in self.sd
                        if self.sd[k].requires_grad and self.sd[k].grad is not None}
        self.opt_d.step()
        for p in self.g_params:
            p.requires_grad_(True)
        return loss_g.detach(), loss_d.detach()
