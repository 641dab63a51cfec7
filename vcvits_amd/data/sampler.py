"""Utterance sharding for data-parallel training (one process per GPU, utterances assigned
rank-strided from a permutation shared by all ranks).  Same index sets as torch.utils.data.DistributedSampler
(what Lightning's `strategy="ddp"` installs for the reference, train.py:99-100), so a run here visits the same
utterances per rank and epoch as the reference would."""
import math

import torch


def rank_indices(n, world_size, rank, epoch=0, seed=0, shuffle=True, drop_last=False):
    """Indices of the utterances rank `rank` of `world_size` processes in epoch `epoch`.

    All ranks draw the same permutation (generator seeded with seed + epoch); the list is padded by wrapping
    around (or truncated with drop_last) to a multiple of world_size and dealt out with stride world_size."""
    if not 0 <= rank < world_size:
        raise ValueError("rank_indices: rank %d outside world of %d" % (rank, world_size))
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if drop_last and n % world_size:
        per = math.ceil((n - world_size) / world_size)
    else:
        per = math.ceil(n / world_size)
    total = per * world_size
    if not drop_last:
        pad = total - len(idx)
        if pad > 0:
            idx += (idx * math.ceil(pad / max(len(idx), 1)))[:pad]
    else:
        idx = idx[:total]
    return idx[rank:total:world_size]


class DistributedUtteranceSampler(torch.utils.data.Sampler):
    """Sampler form of rank_indices (call set_epoch(e) at the start of every epoch, as with torch's)."""

    def __init__(self, dataset_len, world_size, rank, seed=0, shuffle=True, drop_last=False):
        self.n, self.world_size, self.rank = int(dataset_len), world_size, rank
        self.seed, self.shuffle, self.drop_last, self.epoch = seed, shuffle, drop_last, 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __iter__(self):
        return iter(rank_indices(self.n, self.world_size, self.rank, self.epoch, self.seed, self.shuffle, self.drop_last))

    def __len__(self):
        return len(rank_indices(self.n, self.world_size, self.rank, 0, self.seed, False, self.drop_last))
