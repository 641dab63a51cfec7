"""Synthetic batches of the shapes SURVEY.md section 8d fixes (no dataset / checkpoint is
reachable offline): waveforms uniform(-0.9, 0.9), HuBERT features N(0,1), pitch ids in 1..511,
speaker ids in 0..511, half the utterances full length and half shorter to exercise the masks."""
import torch


def vocoder_batch(batch_size, inter_channels, segment_size=16384, hop=512, seed=1234, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(batch_size, inter_channels, segment_size // hop, generator=g)
    y = torch.rand(batch_size, 1, segment_size, generator=g) * 1.8 - 0.9
    return {"z_slice": z.to(device), "y_wav_values": y.to(device)}


def full_batch(batch_size, hubert_channels, t_y=384, t_x=204, hop=512, seed=1234, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    ts = t_y * hop
    y = torch.rand(batch_size, 1, ts, generator=g) * 1.8 - 0.9
    y_len = torch.full((batch_size,), ts, dtype=torch.long)
    x_len = torch.full((batch_size,), t_x, dtype=torch.long)
    y_len[batch_size // 2:] = 300 * hop if t_y >= 300 else ts
    x_len[batch_size // 2:] = 160 if t_x >= 160 else t_x
    for i in range(batch_size):
        y[i, :, int(y_len[i]):] = 0.0
    feats = torch.randn(batch_size, hubert_channels, t_x, generator=g)
    pitch = torch.randint(1, 512, (batch_size, t_x), generator=g)
    sid = torch.randint(0, 512, (batch_size,), generator=g)
    b = {"sid": sid, "x_hubert_features_values": feats, "x_hubert_features_lengths": x_len,
         "x_pitch_values": pitch, "x_pitch_lengths": x_len.clone(), "y_wav_values": y, "y_wav_lengths": y_len}
    return {k: v.to(device) for k, v in b.items()}
