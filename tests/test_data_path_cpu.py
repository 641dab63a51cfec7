"""CPU: the data path -- oracle and host mirror against the fixture generated from the
reference's own functions (tools/make_goldens_data.py -> tests/golden/data_path.npz).  Integer / index work:
bit-exact."""
import os
import types

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN_DIR, "data_path.npz"))


def _rows(gold, as_torch):
    rows = []
    for i in range(5):
        r = {k: gold["row%d_%s" % (i, k)] for k in ("sid", "x_wav", "x_pitch", "y_wav")}
        r["sid"] = int(r["sid"])
        if as_torch:
            r = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in r.items()}
        rows.append(r)
    return rows


def test_oracle_matches_reference_fixture(gold):
    from oracle import data_oracle as D
    assert np.array_equal(D.coarse_f0(gold["f0_in"].copy()), gold["f0_coarse"])
    out, order = D.collate(_rows(gold, False))
    assert order == gold["collate_order"].tolist()
    for k in out:
        assert np.array_equal(out[k], gold["collate_" + k]), k
    for p, names in zip(gold["cache_paths_in"], gold["cache_names"]):
        assert "|".join(D.cache_names(str(p), 16000, 22050, 1024, 1024, 512)) == str(names)
    items = [s.split("|") for s in gold["filelist"].tolist()]
    assert ["|".join(i) for i in D.shuffled(items)] == gold["filelist_shuffled"].tolist()


def test_coarse_f0_bit_exact(gold):
    from vcvits_amd.data import coarse_f0
    got = coarse_f0(torch.from_numpy(gold["f0_in"].copy()))
    assert got.dtype == torch.float32 and np.array_equal(got.numpy(), gold["f0_coarse"])
    assert int(got.min()) >= 1 and int(got.max()) <= 511
    assert coarse_f0(torch.zeros(1, 0)).shape == (1, 0)          # empty utterance
    assert torch.equal(coarse_f0(torch.zeros(1, 9)), torch.ones(1, 9))  # all unvoiced -> class 1


def test_collate_schema_and_values(gold):
    from vcvits_amd.data import VoiceConversionMultiSpeakerCollate
    out = VoiceConversionMultiSpeakerCollate()(_rows(gold, True))
    assert list(out) == ["sid", "x_wav_values", "x_wav_lengths", "x_pitch_values", "x_pitch_lengths", "y_wav_values",
                         "y_wav_lengths"]
    for k, v in out.items():
        ref = gold["collate_" + k]
        assert v.dtype == torch.from_numpy(ref).dtype and np.array_equal(v.numpy(), ref), k
    # rows come out by decreasing source length, padding is zero
    assert out["x_wav_lengths"].tolist() == sorted(out["x_wav_lengths"].tolist(), reverse=True)
    assert float(out["y_wav_values"][-1, 0, int(out["y_wav_lengths"][-1]):].abs().sum()) == 0.0
    with pytest.raises(TypeError):  # the reference raises here too (collate.py:128)
        VoiceConversionMultiSpeakerCollate(return_ids=True)(_rows(gold, True))
    assert str(gold["return_ids_error"]) == "TypeError"
    # a single-row batch and the `vits.` import path of the reference
    from vits.data.collate import VoiceConversionMultiSpeakerCollate as ViaShim
    one = ViaShim()(_rows(gold, True)[:1])
    assert one["x_wav_values"].shape[0] == 1 and int(one["x_wav_lengths"][0]) == one["x_wav_values"].shape[2]


def test_dataset_cache_layout_and_shuffle(gold, tmp_path):
    from vcvits_amd.data.dataset import VoiceConversionMultiSpeakerDataset, cache_paths
    hp = types.SimpleNamespace(source_sampling_rate=16000, target_sampling_rate=22050, filter_length=1024, hop_length=256,
                               win_length=1024, num_pitch=512)
    for p, names in zip(gold["cache_paths_in"], gold["cache_names"]):
        got = cache_paths(str(tmp_path), str(p), hp)
        assert [os.path.basename(g) for g in got] == str(names).split("|")
    listing = tmp_path / "filelist.txt"
    listing.write_text("\n".join(gold["filelist"].tolist()) + "\n", encoding="utf-8")
    ds = VoiceConversionMultiSpeakerDataset(str(listing), hp, str(tmp_path))
    assert ["|".join(i) for i in ds.audiopaths] == gold["filelist_shuffled"].tolist()
    with pytest.raises(FileNotFoundError):
        ds[0]
    calls = []

    def load_audio(path, sr):
        calls.append((path, sr))
        return torch.full((sr // 1000,), float(sr))

    def get_pitch(path, n_fft, win, num_pitch, sr):
        return torch.full((1, 3), 7, dtype=torch.long)

    ds2 = VoiceConversionMultiSpeakerDataset(str(listing), hp, str(tmp_path), load_audio=load_audio, get_pitch=get_pitch)
    item = ds2[0]
    assert item["x_wav"].shape == (1, 16) and item["y_wav"].shape == (1, 22) and item["sid"] == int(ds2.audiopaths[0][1])
    n_calls = len(calls)
    again = ds[0]                                   # now served from the cache the hooks filled, by the hook-less dataset
    assert len(calls) == n_calls and torch.equal(again["x_wav"], item["x_wav"]) and torch.equal(again["x_pitch"], item["x_pitch"])
    assert len(ds) == len(gold["filelist"])


def test_infer_length_scale():
    from oracle import data_oracle as D
    from vcvits_amd.data import infer_length_scale
    hp = types.SimpleNamespace(target_sampling_rate=22050, hop_length=256, source_sampling_rate=16000)
    assert infer_length_scale(hp) == D.length_scale(22050, 256, 16000) == (22050 / 256) / 16000


@pytest.mark.parametrize("n,world", [(23, 2), (23, 8), (64, 8), (5, 8), (1, 2), (100, 3)])
def test_rank_sharding_matches_torch_distributed_sampler(n, world):
    """Utterance sharding: same per-rank index lists as torch.utils.data.DistributedSampler -- what the
    reference's Lightning DDP run uses -- for several epochs, with and without shuffle / drop_last; the ranks
    partition the (padded) epoch."""
    from torch.utils.data import DistributedSampler
    from vcvits_amd.data import DistributedUtteranceSampler, rank_indices

    class _DS:
        def __len__(self):
            return n

    for shuffle in (True, False):
        for drop_last in (False, True):
            for epoch in (0, 1, 7):
                seen = []
                for rank in range(world):
                    ref = DistributedSampler(_DS(), num_replicas=world, rank=rank, shuffle=shuffle, seed=1234, drop_last=drop_last)
                    ref.set_epoch(epoch)
                    mine = rank_indices(n, world, rank, epoch=epoch, seed=1234, shuffle=shuffle, drop_last=drop_last)
                    assert mine == list(ref), (shuffle, drop_last, epoch, rank)
                    s = DistributedUtteranceSampler(n, world, rank, seed=1234, shuffle=shuffle, drop_last=drop_last)
                    s.set_epoch(epoch)
                    assert list(s) == mine and len(s) == len(ref)
                    seen += mine
                if not drop_last:
                    assert set(seen) == set(range(n))          # every utterance visited
                assert len(seen) % world == 0                   # equal work per rank
    with pytest.raises(ValueError):
        rank_indices(10, 2, 2)
