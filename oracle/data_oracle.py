"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's data-path arithmetic in
NumPy / pure Python, each function citing the reference lines it follows.  Only tests/ may import this file.
Pinned against the reference's own functions by tools/make_goldens_data.py (tests/golden/data_path.npz)."""
import hashlib
import random

import numpy as np


def coarse_f0(f0, f0_min=50.0, f0_max=1100.0, f0_bin=512):
    """vits/data/audio.py:65-76 (float32 arithmetic as torch does it)."""
    f0 = np.asarray(f0, dtype=np.float32)
    mel_min = 1127 * np.log(1 + f0_min / 700)
    mel_max = 1127 * np.log(1 + f0_max / 700)
    mel = (np.float32(1127) * np.log(np.float32(1) + f0 / np.float32(700))).astype(np.float32)
    pos = mel > 0
    mel[pos] = ((mel[pos] - np.float32(mel_min)) * np.float32(f0_bin - 2) / np.float32(mel_max - mel_min) + np.float32(1)).astype(np.float32)
    mel[mel <= 1] = 1
    mel[mel > f0_bin - 1] = f0_bin - 1
    return np.round(mel)  # half-to-even, as torch.round


def collate(batch):
    """vits/data/collate.py:137-191: rows sorted by decreasing x_wav length (stable order of torch.sort is
    taken from the golden fixture's `order`), right zero padding."""
    lens = np.array([r["x_wav"].shape[1] for r in batch])
    order = sorted(range(len(batch)), key=lambda i: -lens[i])  # ties: see test (fixture has no ties)
    n = len(batch)
    out = {
        "sid": np.zeros(n, np.int64),
        "x_wav_values": np.zeros((n, 1, max(r["x_wav"].shape[1] for r in batch)), np.float32),
        "x_wav_lengths": np.zeros(n, np.int64),
        "x_pitch_values": np.zeros((n, max(r["x_pitch"].shape[1] for r in batch)), np.int64),
        "x_pitch_lengths": np.zeros(n, np.int64),
        "y_wav_values": np.zeros((n, 1, max(r["y_wav"].shape[1] for r in batch)), np.float32),
        "y_wav_lengths": np.zeros(n, np.int64),
    }
    for i, src in enumerate(order):
        r = batch[src]
        out["sid"][i] = r["sid"]
        out["x_wav_values"][i, :, :r["x_wav"].shape[1]] = r["x_wav"]
        out["x_wav_lengths"][i] = r["x_wav"].shape[1]
        out["x_pitch_values"][i, :r["x_pitch"].shape[1]] = r["x_pitch"][0]
        out["x_pitch_lengths"][i] = r["x_pitch"].shape[1]
        out["y_wav_values"][i, :, :r["y_wav"].shape[1]] = r["y_wav"]
        out["y_wav_lengths"][i] = r["y_wav"].shape[1]
    return out, order


def hash_string(s):
    """vits/data/dataset/vc_ms.py:24-26."""
    return hashlib.md5(s.encode("utf-8")).hexdigest()


def cache_names(audiopath, source_sr, target_sr, filter_length, win_length, num_pitch):
    """vc_ms.py:53, 62-66, 81."""
    return (hash_string(f"{audiopath}_{source_sr}") + ".pt",
            hash_string(f"{audiopath}_{filter_length}_{win_length}_{num_pitch}_{source_sr}") + ".pt",
            hash_string(f"{audiopath}_{target_sr}") + ".pt")


def shuffled(items):
    """vc_ms.py:40-41: random.seed(1234); random.shuffle(list)."""
    items = list(items)
    random.seed(1234)
    random.shuffle(items)
    return items


def length_scale(target_sr, hop_length, source_sr):
    """infer.py:81."""
    return (target_sr / hop_length) / source_sr
