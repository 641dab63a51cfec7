"""CPU oracle for the vcvits hot path -- TEST INFRASTRUCTURE ONLY.

A functional, torch-CPU restatement of the reference's algorithm (vtuber-plan/vcvits), written
against a flat ``state_dict`` whose keys are the reference's own parameter names.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this file; the product path (vcvits_amd/) never does -- it fails loudly without the HIP library.

Pinning (see tools/make_goldens.py and tests/test_oracle_vs_golden.py): every function below was
checked, in the authoring container, against the reference modules imported from /root/reference
on seeded inputs; the captured input/output vectors are committed under tests/golden/.  Three
pieces restate third-party arithmetic that is NOT in the reference tree and therefore stay
"parity unpinned" at that boundary:
  * the HiFi-GAN Generator (vtuber-plan/hifi-gan v0.3.1, call site synthesizer_svc.py:59) --
    restated from the canonical VITS/HiFi-GAN definition using the in-tree ResBlock1
    (modules.py:186-222), get_padding/init_weights (commons.py:8-15), the ctor call
    synthesizer_tts.py:71-78 and configs/base.json:55-63;
  * librosa.filters.mel (librosa 0.10.0.post2, call sites mel_processing.py:103,126);
  * torchaudio.functional.spectrogram (torchaudio 2.0.1, call site mel_processing.py:90) --
    constant pad + torch.stft; checked here against torch.stft only.

Every function cites the reference file:line it follows.  Citations are relative to
/root/reference/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # vits/model/modules.py:16


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def get_padding(kernel_size, dilation=1):
    """vits/commons.py:14-15"""
    return int((kernel_size * dilation - dilation) / 2)


def sequence_mask(length, max_length=None):
    """vits/commons.py:120-124"""
    if max_length is None:
        max_length = int(length.max())
    ar = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return ar.unsqueeze(0) < length.unsqueeze(1)


def slice_segments(x, ids_str, segment_size=4):
    """vits/commons.py:48-54"""
    out = torch.zeros_like(x[:, :, :segment_size])
    for i in range(x.size(0)):
        s = int(ids_str[i])
        out[i] = x[i, :, s:s + segment_size]
    return out


def slice_ids_from_uniform(u, x_lengths, segment_size):
    """vits/commons.py:57-64 with the torch.rand draw `u` [B] made an explicit input."""
    ids_str_max = x_lengths - segment_size + 1
    return (u * ids_str_max).to(dtype=torch.long)


def spectral_norm_weight(sd, prefix, training, eps=1e-12):
    """Effective weight of a conv under torch.nn.utils.spectral_norm (the old hook form; dim 0, one power iteration,
    eps 1e-12), which discriminator.py:17,52 use as norm_f when use_spectral_norm=True.  Restates the published
    algorithm of torch's SpectralNorm.compute_weight: in a training forward  v <- normalize(W^T u),  u <- normalize(W v)
    (both stored vectors updated in place, without gradient), then sigma = u . (W v) on copies of them and W / sigma;
    an eval forward uses the stored vectors as they are."""
    w, u, v = sd[prefix + ".weight_orig"], sd[prefix + ".weight_u"], sd[prefix + ".weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
        u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


def _wn_weight(sd, prefix, training=False):
    """Effective weight of a (possibly weight-normed) conv: torch.nn.utils.weight_norm, dim=0
    (modules.py:126,132,143; discriminator.py:16); spectral norm where the layer carries weight_orig (`training`
    = the module's mode, which decides the power iteration)."""
    if prefix + ".weight_orig" in sd:
        return spectral_norm_weight(sd, prefix, training)
    if prefix + ".weight_v" in sd:
        v, g = sd[prefix + ".weight_v"], sd[prefix + ".weight_g"]
        dims = tuple(range(1, v.dim()))
        return v * (g / v.norm(2, dim=dims, keepdim=True))
    return sd[prefix + ".weight"]


def _bias(sd, prefix):
    return sd.get(prefix + ".bias", None)


# ------------------------------------------------------------------------------------------------
# WN / posterior encoder / flow
# ------------------------------------------------------------------------------------------------
def wn_forward(sd, prefix, x, x_mask, g, hidden, kernel_size, dilation_rate, n_layers):
    """vits/model/modules.py:147-175 (WN.forward); gate = commons.py:99-106."""
    output = torch.zeros_like(x)
    if g is not None:
        g = F.conv1d(g, _wn_weight(sd, prefix + ".cond_layer"), _bias(sd, prefix + ".cond_layer"))
    for i in range(n_layers):
        dil = dilation_rate ** i
        pad = int((kernel_size * dil - dil) / 2)
        x_in = F.conv1d(x, _wn_weight(sd, "%s.in_layers.%d" % (prefix, i)),
                        _bias(sd, "%s.in_layers.%d" % (prefix, i)), dilation=dil, padding=pad)
        if g is not None:
            off = i * 2 * hidden
            in_act = x_in + g[:, off:off + 2 * hidden, :]
        else:
            in_act = x_in
        acts = torch.tanh(in_act[:, :hidden]) * torch.sigmoid(in_act[:, hidden:])
        rs = F.conv1d(acts, _wn_weight(sd, "%s.res_skip_layers.%d" % (prefix, i)),
                      _bias(sd, "%s.res_skip_layers.%d" % (prefix, i)))
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * x_mask


def posterior_encoder_forward(sd, prefix, x, x_lengths, g, eps, out_channels, hidden, kernel_size,
                              dilation_rate, n_layers):
    """vits/model/encoders/posterior_encoder.py:31-39; `eps` is the torch.randn_like draw."""
    x_mask = sequence_mask(x_lengths, x.size(2)).unsqueeze(1).to(x.dtype)
    h = F.conv1d(x, sd[prefix + ".pre.weight"], sd[prefix + ".pre.bias"]) * x_mask
    h = wn_forward(sd, prefix + ".enc", h, x_mask, g, hidden, kernel_size, dilation_rate, n_layers)
    stats = F.conv1d(h, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"]) * x_mask
    m, logs = torch.split(stats, out_channels, dim=1)
    z = (m + eps * torch.exp(logs)) * x_mask
    return z, m, logs, x_mask


def coupling_forward(sd, prefix, x, x_mask, g, reverse, channels, hidden, kernel_size, dilation_rate,
                     n_layers):
    """vits/model/modules.py:317-336 (ResidualCouplingLayer, mean_only=True as flow.py:27)."""
    half = channels // 2
    x0, x1 = x[:, :half], x[:, half:]
    h = F.conv1d(x0, sd[prefix + ".pre.weight"], sd[prefix + ".pre.bias"]) * x_mask
    h = wn_forward(sd, prefix + ".enc", h, x_mask, g, hidden, kernel_size, dilation_rate, n_layers)
    m = F.conv1d(h, sd[prefix + ".post.weight"], sd[prefix + ".post.bias"]) * x_mask
    if not reverse:
        x1 = m + x1 * x_mask
    else:
        x1 = (x1 - m) * x_mask
    return torch.cat([x0, x1], 1)


def flow_forward(sd, prefix, x, x_mask, g, reverse, channels, hidden, kernel_size, dilation_rate,
                 n_layers, n_flows=4):
    """vits/model/flow.py:30-37 + Flip (modules.py:261-268).  Even indices are coupling layers,
    odd indices channel flips."""
    if not reverse:
        for i in range(n_flows):
            x = coupling_forward(sd, "%s.flows.%d" % (prefix, 2 * i), x, x_mask, g, False, channels,
                                 hidden, kernel_size, dilation_rate, n_layers)
            x = torch.flip(x, [1])
    else:
        for i in reversed(range(n_flows)):
            x = torch.flip(x, [1])
            x = coupling_forward(sd, "%s.flows.%d" % (prefix, 2 * i), x, x_mask, g, True, channels,
                                 hidden, kernel_size, dilation_rate, n_layers)
    return x


# ------------------------------------------------------------------------------------------------
# relative-position transformer encoder
# ------------------------------------------------------------------------------------------------
def channel_layer_norm(x, gamma, beta, eps=1e-5):
    """vits/model/modules.py:19-31 (LayerNorm over the channel dim of [B,C,T])."""
    xt = x.transpose(1, -1)
    xt = F.layer_norm(xt, (x.size(1),), gamma, beta, eps)
    return xt.transpose(1, -1)


def rel_attention(sd, prefix, x, attn_mask, n_heads, window_size, drop=None):
    """vits/model/transformer/relative_attention_transformer.py:136-185 (self-attention with
    shared-head windowed relative key/value embeddings), written with explicit band indexing
    instead of the pad/reshape skew (:217-251): rel_logits[b,h,i,j] = q_i . E_k[j-i+w] for
    |j-i| <= w, and out_i += sum_{|j-i|<=w} p[i,j] * E_v[j-i+w].  `drop`: None = dropout is identity (eval); else a
    callable applied to the probabilities where the reference applies `self.drop` (:173), returning them multiplied by
    a dropout mask / (1 - p) -- the tests feed it the masks the HIP step drew, so both sides drop the same elements."""
    q = F.conv1d(x, sd[prefix + ".conv_q.weight"], sd[prefix + ".conv_q.bias"])
    k = F.conv1d(x, sd[prefix + ".conv_k.weight"], sd[prefix + ".conv_k.bias"])
    v = F.conv1d(x, sd[prefix + ".conv_v.weight"], sd[prefix + ".conv_v.bias"])
    b, d, t = q.shape
    dk = d // n_heads
    q = q.view(b, n_heads, dk, t).transpose(2, 3)
    k = k.view(b, n_heads, dk, t).transpose(2, 3)
    v = v.view(b, n_heads, dk, t).transpose(2, 3)
    qs = q / math.sqrt(dk)
    scores = torch.matmul(qs, k.transpose(-2, -1))
    emb_k = sd[prefix + ".emb_rel_k"][0]  # [2w+1, dk], heads_share
    emb_v = sd[prefix + ".emb_rel_v"][0]
    w = window_size
    rel = torch.matmul(qs, emb_k.t())  # [b,h,t,2w+1]
    idx_i = torch.arange(t).unsqueeze(1)
    idx_j = torch.arange(t).unsqueeze(0)
    off = idx_j - idx_i + w  # [t,t]
    band = (off >= 0) & (off <= 2 * w)
    offc = off.clamp(0, 2 * w)
    local = torch.gather(rel, 3, offc.expand(b, n_heads, t, t)) * band
    scores = scores + local
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p = F.softmax(scores, dim=-1)
    if drop is not None:
        p = drop(p)
    out = torch.matmul(p, v)
    # relative values: weights[b,h,i,r] = p[b,h,i,i+r-w]
    pw = torch.zeros(b, n_heads, t, 2 * w + 1, dtype=p.dtype)
    for r in range(2 * w + 1):
        j = torch.arange(t) + r - w
        ok = (j >= 0) & (j < t)
        jj = j.clamp(0, t - 1)
        pw[:, :, :, r] = p[:, :, torch.arange(t), jj] * ok
    out = out + torch.matmul(pw, emb_v)
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    out = F.conv1d(out, sd[prefix + ".conv_o.weight"], sd[prefix + ".conv_o.bias"])
    return out, p


def ffn_forward(sd, prefix, x, x_mask, kernel_size, drop=None):
    """relative_attention_transformer.py:285-311 (non-causal FFN, ReLU, dropout (:304), same padding)."""
    pl, pr = (kernel_size - 1) // 2, kernel_size // 2
    h = F.conv1d(F.pad(x * x_mask, (pl, pr)), sd[prefix + ".conv_1.weight"], sd[prefix + ".conv_1.bias"])
    h = torch.relu(h)
    if drop is not None:
        h = drop(h)
    h = F.conv1d(F.pad(h * x_mask, (pl, pr)), sd[prefix + ".conv_2.weight"], sd[prefix + ".conv_2.bias"])
    return h * x_mask


def transformer_encoder_forward(sd, prefix, x, x_mask, n_heads, n_layers, kernel_size, window_size=4, drop=None):
    """relative_attention_transformer.py:35-47 (post-LN encoder).  drop: see rel_attention (None: identity); called in
    the reference's order per layer: attention probabilities, attention output (:40), FFN hidden (:304), FFN output (:44)."""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    for i in range(n_layers):
        y, _ = rel_attention(sd, "%s.attn_layers.%d" % (prefix, i), x, attn_mask, n_heads, window_size, drop=drop)
        if drop is not None:
            y = drop(y)
        x = channel_layer_norm(x + y, sd["%s.norm_layers_1.%d.gamma" % (prefix, i)],
                               sd["%s.norm_layers_1.%d.beta" % (prefix, i)])
        y = ffn_forward(sd, "%s.ffn_layers.%d" % (prefix, i), x, x_mask, kernel_size, drop=drop)
        if drop is not None:
            y = drop(y)
        x = channel_layer_norm(x + y, sd["%s.norm_layers_2.%d.gamma" % (prefix, i)],
                               sd["%s.norm_layers_2.%d.beta" % (prefix, i)])
    return x * x_mask


def content_encoder_forward(sd, prefix, feats, x_lengths, pitch, out_channels, n_heads, n_layers,
                            kernel_size, preload=False, drop=None):
    """Post-HuBERT half of HubertContentEncoder.forward (content_encoder.py:58-73): `feats`
    [B, hubert, T] stands for x_encoded.  preload=True follows PreloadHubertContentEncoder
    (content_encoder.py:110-126: half-width projections concatenated)."""
    hub = F.linear(feats.transpose(1, -1), sd[prefix + ".hubert_proj.weight"],
                   sd[prefix + ".hubert_proj.bias"]).transpose(1, -1)
    pe = F.embedding(pitch, sd[prefix + ".emb_pitch.weight"]).transpose(1, -1)
    if preload:
        pe = F.linear(pe.transpose(1, -1), sd[prefix + ".pitch_proj.weight"],
                      sd[prefix + ".pitch_proj.bias"]).transpose(1, -1)
        out = torch.cat((hub, pe), dim=1)
    else:
        out = hub + pe
    x_mask = sequence_mask(x_lengths.int(), out.size(2)).unsqueeze(1).to(feats.dtype)
    x_out = transformer_encoder_forward(sd, prefix + ".encoder", out * x_mask, x_mask, n_heads, n_layers,
                                        kernel_size, drop=drop)
    stats = F.conv1d(x_out, sd[prefix + ".proj.weight"], sd[prefix + ".proj.bias"]) * x_mask
    m, logs = torch.split(stats, out_channels, dim=1)
    return x_out, m, logs, x_mask


# ------------------------------------------------------------------------------------------------
# HiFi-GAN generator (canonical definition; parity unpinned -- see module docstring)
# ------------------------------------------------------------------------------------------------
def resblock1_forward(sd, prefix, x, kernel_size, dilations=(1, 3, 5)):
    """vits/model/modules.py:203-216 (ResBlock1.forward, x_mask=None)."""
    for i, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _wn_weight(sd, "%s.convs1.%d" % (prefix, i)), _bias(sd, "%s.convs1.%d" % (prefix, i)),
                      dilation=d, padding=get_padding(kernel_size, d))
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, _wn_weight(sd, "%s.convs2.%d" % (prefix, i)), _bias(sd, "%s.convs2.%d" % (prefix, i)),
                      dilation=1, padding=get_padding(kernel_size, 1))
        x = xt + x
    return x


def resblock2_forward(sd, prefix, x, kernel_size, dilations=(1, 3)):
    """vits/model/modules.py:234-243 (ResBlock2.forward, x_mask=None): one weight-normed dilated conv per residual step."""
    for i, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _wn_weight(sd, "%s.convs.%d" % (prefix, i)), _bias(sd, "%s.convs.%d" % (prefix, i)),
                      dilation=d, padding=get_padding(kernel_size, d))
        x = xt + x
    return x


def generator_forward(sd, prefix, x, upsample_rates=(8, 8, 4, 2), upsample_kernel_sizes=(16, 16, 4, 4),
                      resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5),) * 3, resblock="1"):
    """conv_pre k7 -> 4 x (leaky 0.1 -> weight-normed ConvTranspose1d ->
    mean of 3 ResBlock1 (resblock="2": ResBlock2, hifi-gan config_v3 style)) -> leaky (default slope 0.01) -> conv_post k7 (no bias) -> tanh.  conv_pre / conv_post are
    read in whichever form the state_dict holds them (the original HiFi-GAN lineage weight-norms both and keeps
    conv_post's bias; the VITS lineage does neither)."""
    nk = len(resblock_kernel_sizes)
    x = F.conv1d(x, _wn_weight(sd, prefix + ".conv_pre"), sd[prefix + ".conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, _wn_weight(sd, "%s.ups.%d" % (prefix, i)), _bias(sd, "%s.ups.%d" % (prefix, i)),
                               stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            rb = resblock1_forward if str(resblock) == "1" else resblock2_forward
            r = rb(sd, "%s.resblocks.%d" % (prefix, i * nk + j), x, resblock_kernel_sizes[j], resblock_dilation_sizes[j])
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, _wn_weight(sd, prefix + ".conv_post"), _bias(sd, prefix + ".conv_post"), padding=3)
    return torch.tanh(x)


# ------------------------------------------------------------------------------------------------
# discriminators
# ------------------------------------------------------------------------------------------------
DISC_S_LAYERS = [  # (stride, padding, groups)  vits/model/discriminators/discriminator.py:53-61
    (1, 7, 1), (4, 20, 4), (4, 20, 16), (4, 20, 64), (4, 20, 256), (1, 2, 1)]


def disc_s_forward(sd, prefix, x, training=False):
    """discriminator.py:63-74 (DiscriminatorS.forward).  `training` matters under spectral norm only."""
    fmap = []
    for i, (s, p, g) in enumerate(DISC_S_LAYERS):
        x = F.conv1d(x, _wn_weight(sd, "%s.convs.%d" % (prefix, i), training), _bias(sd, "%s.convs.%d" % (prefix, i)),
                     stride=s, padding=p, groups=g)
        x = F.leaky_relu(x, LRELU_SLOPE)
        fmap.append(x)
    x = F.conv1d(x, _wn_weight(sd, prefix + ".conv_post", training), _bias(sd, prefix + ".conv_post"), padding=1)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def disc_p_forward(sd, prefix, x, period, kernel_size=5, stride=3, training=False):
    """discriminator.py:27-46 (DiscriminatorP.forward).  `training` matters under spectral norm only."""
    fmap = []
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), "reflect")
        t = t + n_pad
    x = x.view(b, c, t // period, period)
    for i in range(5):
        s = stride if i < 4 else 1
        x = F.conv2d(x, _wn_weight(sd, "%s.convs.%d" % (prefix, i), training), _bias(sd, "%s.convs.%d" % (prefix, i)),
                     stride=(s, 1), padding=(get_padding(kernel_size, 1), 0))
        x = F.leaky_relu(x, LRELU_SLOPE)
        fmap.append(x)
    x = F.conv2d(x, _wn_weight(sd, prefix + ".conv_post", training), _bias(sd, prefix + ".conv_post"), padding=(1, 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def mpd_forward(sd, prefix, y, y_hat, periods, training=False):
    """multi_period_discriminator.py:17-31: one DiscriminatorS followed by one DiscriminatorP per
    period; d(y) then d(y_hat) -- two forwards, i.e. two power iterations per layer under spectral norm."""
    outs = ([], [], [], [])
    for i in range(len(periods) + 1):
        p = "%s.discriminators.%d" % (prefix, i)
        if i == 0:
            r, fr = disc_s_forward(sd, p, y, training)
            g, fg = disc_s_forward(sd, p, y_hat, training)
        else:
            r, fr = disc_p_forward(sd, p, y, periods[i - 1], training=training)
            g, fg = disc_p_forward(sd, p, y_hat, periods[i - 1], training=training)
        outs[0].append(r); outs[1].append(g); outs[2].append(fr); outs[3].append(fg)
    return outs


def msd_forward(sd, prefix, y, y_hat, training=False):
    """multi_scale_discriminator.py:27-42: 5 DiscriminatorS on progressively AvgPool1d(4,2,2)
    inputs."""
    outs = ([], [], [], [])
    for i in range(5):
        if i != 0:
            y = F.avg_pool1d(y, 4, 2, 2)
            y_hat = F.avg_pool1d(y_hat, 4, 2, 2)
        p = "%s.discriminators.%d" % (prefix, i)
        r, fr = disc_s_forward(sd, p, y, training)
        g, fg = disc_s_forward(sd, p, y_hat, training)
        outs[0].append(r); outs[2].append(fr); outs[1].append(g); outs[3].append(fg)
    return outs


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
def _f(t):
    """`.float()` of the reference's losses (losses.py upcasts fp16 autocast outputs); a float64 run of the oracle -- the
    yardstick of the fp64 ranking tests -- stays float64."""
    return t if t.dtype == torch.float64 else t.float()


def feature_loss(fmap_r, fmap_g):
    """vits/light/losses.py:4-12"""
    loss = 0
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean(torch.abs(_f(rl).detach() - _f(gl)))
    return loss * 2


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    """vits/light/losses.py:14-27 (without the .item() logging lists)."""
    loss = 0
    for dr, dg in zip(disc_real_outputs, disc_generated_outputs):
        loss = loss + torch.mean((1 - _f(dr)) ** 2) + torch.mean(_f(dg) ** 2)
    return loss


def generator_loss(disc_outputs):
    """vits/light/losses.py:29-38"""
    loss = 0
    for dg in disc_outputs:
        loss = loss + torch.mean((1 - _f(dg)) ** 2)
    return loss


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    """vits/light/losses.py:40-55"""
    kl = logs_p - logs_q - 0.5
    kl = kl + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2. * logs_p)
    kl = torch.sum(kl * z_mask)
    return kl / torch.sum(z_mask)


# ------------------------------------------------------------------------------------------------
# STFT / mel
# ------------------------------------------------------------------------------------------------
def spectrogram(y, n_fft, hop_size, win_size, reflect):
    """reflect=True: spectrogram_torch (mel_processing.py:54-74); reflect=False:
    spectrogram_torch_audio (:76-96) = torchaudio.functional.spectrogram(pad=(n_fft-hop)/2), which
    zero-pads.  Returns sqrt(re^2 + im^2 + 1e-6)."""
    pad = int((n_fft - hop_size) / 2)
    win = torch.hann_window(win_size, dtype=y.dtype)
    lead = y.shape[:-1]
    y2 = y.reshape(-1, y.shape[-1])
    yp = F.pad(y2.unsqueeze(1), (pad, pad), mode="reflect" if reflect else "constant").squeeze(1)
    spec = torch.stft(yp, n_fft, hop_length=hop_size, win_length=win_size, window=win, center=False,
                      normalized=False, onesided=True, return_complex=True)
    mag = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-6)
    return mag.reshape(lead + mag.shape[-2:])


def mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """librosa.filters.mel (htk=False, norm='slaney', float32) -- SURVEY.md Appendix B."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)

    def hz_to_mel(f):
        f = np.asanyarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asanyarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        freqs = f_sp * m
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)

    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


def spec_to_mel(spec, melmat):
    """mel_processing.py:98-112: log(clamp(M @ spec, 1e-5))."""
    return torch.log(torch.clamp(torch.matmul(melmat, spec), min=1e-5))


def audio_pipeline(waveform, n_fft=2048, hop_length=512, win_length=2048):
    """vits/model/pipeline.py:49-70 (aug=False): torchaudio Spectrogram(pad=(n_fft-hop)/2, power=None,
    center=False) -> InverseSpectrogram -> copied into zeros_like(waveform).  torchaudio (2.0.1, absent) is restated
    per SURVEY Appendix C: constant pad + torch.stft; InverseSpectrogram = torch.istft(center=True)."""
    b, c, t = waveform.shape
    pad = int((n_fft - hop_length) / 2)
    win = torch.hann_window(win_length)
    yp = F.pad(waveform.reshape(b * c, t), (pad, pad))
    spec = torch.stft(yp, n_fft, hop_length=hop_length, win_length=win_length, window=win, center=False,
                      normalized=False, onesided=True, return_complex=True)
    wav = torch.istft(spec, n_fft, hop_length=hop_length, win_length=win_length, window=win, center=True,
                      normalized=False, onesided=True).reshape(b, c, -1)
    out = torch.zeros_like(waveform)
    n = min(wav.shape[2], t)
    out[:, :, :n] = wav[:, :, :n]
    return out


def hubert_features(extractor, x_wav):
    """content_encoder.py:53-56: pad (400 - 320) // 2 samples each side, `hubert.extract_features(wav.squeeze(1))`
    -> [B, T', H], transposed to [B, H, T'] (the reference transposes twice around hubert_proj: a no-op pair)."""
    wav = F.pad(x_wav, ((400 - 320) // 2, (400 - 320) // 2))
    x_encoded, _ = extractor.extract_features(wav.squeeze(1))
    return x_encoded.transpose(1, -1)
