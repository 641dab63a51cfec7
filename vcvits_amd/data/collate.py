"""Batch schema of the voice-conversion trainer (reference: vits/data/collate.py:133-191)."""
import torch


class VoiceConversionMultiSpeakerCollate:
    """Rows {"sid", "x_wav" [1, Tx], "x_pitch" [1, Tp], "y_wav" [1, Ty]} -> right-zero-padded batch, rows
    ordered by DECREASING source length (torch.sort(descending=True) on the x_wav lengths: ties keep
    torch's sort order).  Keys and dtypes as the reference: sid / *_lengths int64, *_wav_values float32
    [B, 1, T], x_pitch_values int64 [B, Tp].

    return_ids=True fails in the reference as well (collate.py:128: `dict.update("ids", ...)` is a
    TypeError); the same exception type is raised here so callers see identical behaviour."""

    def __init__(self, return_ids: bool = False):
        self.return_ids = return_ids

    def __call__(self, batch):
        n = len(batch)
        x_len = torch.tensor([row["x_wav"].size(1) for row in batch], dtype=torch.long)
        _, order = torch.sort(x_len, dim=0, descending=True)
        max_x = max(row["x_wav"].size(1) for row in batch)
        max_p = max(row["x_pitch"].size(1) for row in batch)
        max_y = max(row["y_wav"].size(1) for row in batch)
        out = {
            "sid": torch.zeros(n, dtype=torch.long),
            "x_wav_values": torch.zeros(n, 1, max_x, dtype=torch.float32),
            "x_wav_lengths": torch.zeros(n, dtype=torch.long),
            "x_pitch_values": torch.zeros(n, max_p, dtype=torch.long),
            "x_pitch_lengths": torch.zeros(n, dtype=torch.long),
            "y_wav_values": torch.zeros(n, 1, max_y, dtype=torch.float32),
            "y_wav_lengths": torch.zeros(n, dtype=torch.long),
        }
        for i, src in enumerate(order.tolist()):
            row = batch[src]
            out["sid"][i] = int(row["sid"])
            for key, name in (("x_wav", "x_wav"), ("y_wav", "y_wav")):
                t = row[key]
                out[name + "_values"][i, :, :t.size(1)] = t
                out[name + "_lengths"][i] = t.size(1)
            pitch = row["x_pitch"]
            out["x_pitch_values"][i, :pitch.size(1)] = pitch[0] if pitch.dim() == 2 else pitch
            out["x_pitch_lengths"][i] = pitch.size(1)
        if self.return_ids:
            raise TypeError("update expected at most 1 argument, got 2")  # collate.py:128 of the reference
        return out
