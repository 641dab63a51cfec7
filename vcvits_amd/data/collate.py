"""Batch schema of the voice-conversion trainer (reference: vits/data/collate.py:133-191), plus length bucketing: the
reference pads every batch to ITS OWN longest utterance, so two batches almost never share a shape; a recorded training
batch (light/graphed.py) replays only when shapes repeat.  `pad_multiple` / `bucket_batch` round the padded lengths UP to
multiples of a bucket (more right-zero padding, same `*_lengths`): every consumer of the padded tensors masks by the
lengths (sequence_mask in the encoders / flow, rand_slice_segments draws inside the lengths), so the valid positions
compute what they compute in the reference's padding -- tests/test_length_buckets_gpu.py holds the two against each other."""
import torch


def round_up(n, m):
    return n if not m or m <= 1 else ((int(n) + m - 1) // m) * m


def bucket_multiples(hop_length, frames=64):
    """Bucket sizes of a config: `frames` spectrogram frames of target waveform (hop_length samples each) and `frames`
    content frames of precomputed features / pitch ids.  frames = 64 gives the 384-frame / 204-frame synthetic utterance of
    SURVEY 8d at most 6 x 4 shapes."""
    return {"y_wav": frames * hop_length, "x_pitch": frames, "x_hubert_features": frames, "noise": frames}


def bucket_batch(batch, multiples, hop_length):
    """Right-zero-pad the `<name>_values` tensors of a collated batch (CPU or GPU) along their last dimension up to the
    next multiple of multiples[name]; `noise` (an injected posterior draw, [B, C, frames]) follows the spectrogram frames.
    Lengths, ids and everything unnamed pass through.  Returns a new dict (tensors already at a multiple are shared).

    One thing in the model depends on the PADDED sizes themselves: synthesizer_svc.py:82-83 stretches the prior statistics
    from the padded content length onto the padded spectrogram length with F.interpolate(mode="nearest").  The batch
    therefore carries `bucket_raw_sizes` = int64 [content frames (0: not re-padded), spectrogram frames] of the padding it
    ARRIVED with, and SynthesizerSVC.forward interpolates with that map (ops.interpolate_nearest(raw_sizes=...)): the valid
    positions get exactly the alignment the reference's collate gives them.  Content-side tensors are bucketed only for
    batches of precomputed features (`x_hubert_features_values`): the frame count behind a padded SOURCE WAVEFORM is the
    feature extractor's business, so `x_wav_values` and its pitch track stay as collated."""
    out = dict(batch)
    feats = batch.get("x_hubert_features_values")
    y = batch.get("y_wav_values")
    if y is not None and "bucket_raw_sizes" not in batch:
        out["bucket_raw_sizes"] = torch.tensor([feats.shape[-1] if feats is not None else 0, y.shape[-1] // int(hop_length)],
                                               dtype=torch.int64, device=y.device)
    for name, mult in multiples.items():
        if name in ("x_pitch", "x_hubert_features") and feats is None:
            continue
        key = name + "_values" if name + "_values" in batch else name
        t = batch.get(key)
        if t is None or not torch.is_tensor(t) or t.dim() < 2:
            continue
        n = t.shape[-1]
        want = round_up(n, mult)
        if want != n:
            out[key] = torch.nn.functional.pad(t, (0, want - n))
    return out


class VoiceConversionMultiSpeakerCollate:
    """Rows {"sid", "x_wav" [1, Tx], "x_pitch" [1, Tp], "y_wav" [1, Ty]} -> right-zero-padded batch, rows
    ordered by DECREASING source length (torch.sort(descending=True) on the x_wav lengths: ties keep
    torch's sort order).  Keys and dtypes as the reference: sid / *_lengths int64, *_wav_values float32
    [B, 1, T], x_pitch_values int64 [B, Tp].

    return_ids=True fails in the reference as well (collate.py:128: `dict.update("ids", ...)` is a
    TypeError); the same exception type is raised here so callers see identical behaviour."""

    def __init__(self, return_ids: bool = False, bucket_frames: int = 0, hop_length: int = 0):
        """bucket_frames = 0: the reference (pad to the batch's longest).  bucket_frames = N with the config's hop_length:
        the target waveform is padded to a multiple of N spectrogram frames so that batch shapes repeat, and the batch carries
        `bucket_raw_sizes` (see bucket_batch); the source waveform and its pitch track stay as the reference pads them."""
        self.return_ids = return_ids
        self.bucket_frames, self.hop_length = int(bucket_frames), int(hop_length)
        if self.bucket_frames > 0 and self.hop_length <= 0:
            raise ValueError("VoiceConversionMultiSpeakerCollate: bucket_frames needs the config's hop_length")

    def __call__(self, batch):
        n = len(batch)
        x_len = torch.tensor([row["x_wav"].size(1) for row in batch], dtype=torch.long)
        _, order = torch.sort(x_len, dim=0, descending=True)
        max_x = max(row["x_wav"].size(1) for row in batch)
        max_p = max(row["x_pitch"].size(1) for row in batch)
        raw_y = max(row["y_wav"].size(1) for row in batch)
        max_y = round_up(raw_y, self.bucket_frames * self.hop_length)
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
        if self.bucket_frames > 0:
            out["bucket_raw_sizes"] = torch.tensor([0, raw_y // self.hop_length], dtype=torch.int64)
        return out
