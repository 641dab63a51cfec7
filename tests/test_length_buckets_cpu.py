"""CPU: length bucketing of collated batches (vcvits_amd/data/collate.py).  The reference collate (vits/data/collate.py:133-190)
pads to each batch's own maximum; bucketing only ADDS right-zero padding up to a multiple, lengths untouched, and records
the sizes the batch arrived with (the prior statistics are stretched over PADDED sizes: synthesizer_svc.py:82-83)."""
import torch

from vcvits_amd.data.collate import (VoiceConversionMultiSpeakerCollate, bucket_batch, bucket_multiples, round_up)

HOP = 512


def _rows(seed, n=5):
    g = torch.Generator().manual_seed(seed)
    rows = []
    for i in range(n):
        tx = int(torch.randint(3000, 9000, (1,), generator=g))
        ty = int(torch.randint(5000, 20000, (1,), generator=g))
        rows.append({"sid": i, "x_wav": torch.randn(1, tx, generator=g), "x_pitch": torch.randint(1, 255, (1, tx // 320), generator=g),
                     "y_wav": torch.randn(1, ty, generator=g)})
    return rows


def test_bucketed_collate_is_the_reference_collate_plus_zero_padding_of_the_target():
    rows = _rows(1)
    ref = VoiceConversionMultiSpeakerCollate()(rows)
    out = VoiceConversionMultiSpeakerCollate(bucket_frames=8, hop_length=HOP)(rows)
    assert set(out) == set(ref) | {"bucket_raw_sizes"}
    for k in ref:
        if k == "y_wav_values":
            n = ref[k].shape[-1]
            assert out[k].shape[-1] == round_up(n, 8 * HOP) and out[k].shape[:-1] == ref[k].shape[:-1]
            assert torch.equal(out[k][..., :n], ref[k]) and not out[k][..., n:].any()
        else:  # lengths, ids, the source waveform and its pitch track: as the reference collates them
            assert torch.equal(out[k], ref[k]) and out[k].dtype == ref[k].dtype, k
    assert out["bucket_raw_sizes"].tolist() == [0, ref["y_wav_values"].shape[-1] // HOP]
    assert "bucket_raw_sizes" not in ref


def _feature_batch(seed, B=4):
    g = torch.Generator().manual_seed(seed)
    ty = int(torch.randint(200, 384, (1,), generator=g))
    tx = int(ty * 0.53)
    return {"sid": torch.arange(B), "x_hubert_features_values": torch.randn(B, 6, tx, generator=g),
            "x_hubert_features_lengths": torch.full((B,), tx), "x_pitch_values": torch.randint(1, 255, (B, tx), generator=g),
            "x_pitch_lengths": torch.full((B,), tx), "y_wav_values": torch.randn(B, 1, ty * HOP, generator=g),
            "y_wav_lengths": torch.full((B,), ty * HOP), "noise": torch.randn(B, 4, ty, generator=g), "ids_slice": torch.zeros(B)}


def test_bucket_batch_pads_every_frame_axis_of_a_feature_batch_and_shapes_repeat():
    mult = bucket_multiples(HOP, 64)
    shapes, raw = set(), set()
    for seed in range(40):
        b = _feature_batch(seed)
        out = bucket_batch(b, mult, HOP)
        tx, ty = b["x_pitch_values"].shape[-1], b["noise"].shape[-1]
        assert out["bucket_raw_sizes"].tolist() == [tx, ty]
        for k, n, m in (("x_hubert_features_values", tx, 64), ("x_pitch_values", tx, 64), ("noise", ty, 64),
                        ("y_wav_values", ty * HOP, 64 * HOP)):
            assert out[k].shape[-1] == round_up(n, m) and torch.equal(out[k][..., :n], b[k]) and not out[k][..., n:].any(), k
        for k in ("sid", "x_hubert_features_lengths", "x_pitch_lengths", "y_wav_lengths", "ids_slice"):
            assert out[k] is b[k]
        shapes.add(tuple(tuple(v.shape) for _, v in sorted(out.items())))
        raw.add(tuple(tuple(v.shape) for _, v in sorted(b.items())))
    assert len(raw) >= 30 and len(shapes) <= 6, (len(raw), len(shapes))
    again = bucket_batch(out, mult, HOP)  # idempotent: the recorded raw sizes are the ORIGINAL ones
    assert again["bucket_raw_sizes"] is out["bucket_raw_sizes"] and again["y_wav_values"] is out["y_wav_values"]


def test_bucket_batch_leaves_the_source_side_of_a_waveform_batch_alone():
    b = VoiceConversionMultiSpeakerCollate()(_rows(3))
    out = bucket_batch(b, bucket_multiples(HOP, 8), HOP)
    assert out["x_wav_values"] is b["x_wav_values"] and out["x_pitch_values"] is b["x_pitch_values"]
    assert out["y_wav_values"].shape[-1] == round_up(b["y_wav_values"].shape[-1], 8 * HOP)
    assert out["bucket_raw_sizes"].tolist() == [0, b["y_wav_values"].shape[-1] // HOP]
