"""Dataset over the reference's pre-processed cache (reference: vits/data/dataset/vc_ms.py:24-112).

The reference decodes / resamples audio (torchaudio) and tracks pitch (librosa pYIN) on a cache miss and
stores three tensors per utterance under md5-derived names.  This mirror reads that cache layout; on a miss
it calls the `decode` hooks the caller supplies (none are bundled: the third-party stack is not part of this
build) and stores the result under the same names, so caches are interchangeable with the reference's."""
import hashlib
import os
import random

import torch


def hash_string(s: str) -> str:
    return hashlib.md5(s.encode("utf-8")).hexdigest()


def cache_paths(cache_dir, audiopath, hp):
    """The three cache files of one utterance (vc_ms.py:53,62-66,81): source-rate wav, pitch classes, target-rate wav."""
    x = hash_string("%s_%s" % (audiopath, hp.source_sampling_rate))
    p = hash_string("%s_%s_%s_%s_%s" % (audiopath, hp.filter_length, hp.win_length, hp.num_pitch, hp.source_sampling_rate))
    y = hash_string("%s_%s" % (audiopath, hp.target_sampling_rate))
    return tuple(os.path.join(cache_dir, h + ".pt") for h in (x, p, y))


def load_filepaths_and_text(filename, split="|"):
    with open(filename, encoding="utf-8") as f:
        return [line.strip().split(split) for line in f]


class VoiceConversionMultiSpeakerDataset(torch.utils.data.Dataset):
    """Items {"sid", "x_wav" [1, Tx] (source rate), "x_pitch" (pitch classes), "y_wav" [1, Ty] (target rate)}.
    The file list is shuffled once with random.seed(1234), as the reference does at construction."""

    def __init__(self, audiopaths, hparams, cache_dir, load_audio=None, get_pitch=None):
        self.audiopaths = load_filepaths_and_text(audiopaths) if isinstance(audiopaths, str) else [list(a) for a in audiopaths]
        self.hparams = hparams
        self.cache_dir = cache_dir
        self._load_audio, self._get_pitch = load_audio, get_pitch
        random.seed(1234)
        random.shuffle(self.audiopaths)

    def _cached(self, path, make):
        if os.path.exists(path):
            return torch.load(path)
        if make is None:
            raise FileNotFoundError("%s is not in the cache and no decoder was supplied (audio decoding / pYIN are "
                                    "third-party steps of the reference, not part of this build)" % path)
        value = make()
        torch.save(value, path)
        return value

    def get_item(self, index):
        item = self.audiopaths[index]
        audiopath = item[0]
        sid = 0 if len(item) == 1 else int(item[1])
        hp = self.hparams
        xp, pp, yp = cache_paths(self.cache_dir, audiopath, hp)
        la, gp = self._load_audio, self._get_pitch
        x_wav = self._cached(xp, None if la is None else (lambda: la(audiopath, sr=hp.source_sampling_rate).unsqueeze(0)))
        x_pitch = self._cached(pp, None if gp is None else (lambda: gp(audiopath, hp.filter_length, hp.win_length,
                                                                        hp.num_pitch, hp.source_sampling_rate)))
        y_wav = self._cached(yp, None if la is None else (lambda: la(audiopath, sr=hp.target_sampling_rate).unsqueeze(0)))
        return {"sid": sid, "x_wav": x_wav, "x_pitch": x_pitch, "y_wav": y_wav}

    def __getitem__(self, index):
        return self.get_item(index)

    def __len__(self):
        return len(self.audiopaths)
