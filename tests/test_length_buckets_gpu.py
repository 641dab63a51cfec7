"""GPU: a stream of VARIABLE-LENGTH batches through the recorded training batch (light/graphed.py).  The reference collate
(vits/data/collate.py:133-190) pads each batch to its own longest utterance, so raw shapes never repeat and nothing would
ever replay; with `train.length_bucket_frames` the module rounds the padded lengths up to bucket multiples
(data/collate.py: bucket_batch), shapes repeat, and the batches run as HIP-graph replays.  Held against the eager loop on
the UN-bucketed batches: the extra right-zero padding must not change what the valid positions compute."""
import copy
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

IN_CHILD = os.environ.get("VCVITS_BUCKET_TESTS_CHILD") == "1"
in_child = pytest.mark.skipif(not IN_CHILD, reason="runs inside test_bucket_tests_in_child_process")


def test_bucket_tests_in_child_process(gpu):
    """(a GPU fault in a graph replay must not take the session down: same arrangement as tests/test_graphed_gpu.py)"""
    if IN_CHILD:
        pytest.skip("this is the child")
    env = dict(os.environ, VCVITS_BUCKET_TESTS_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "variable_length"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    assert "2 passed" in out, out[-1500:]


def _cfg(bucket):
    from vcvits_amd import configs
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 32, "hidden_channels": 32, "filter_channels": 64, "n_heads": 2,
                         "upsample_initial_channel": 64, "hubert_channels": 48, "gin_channels": 16, "p_dropout": 0.0,
                         "multi_period_discriminator_periods": [2, 3]})
    cfg["data"]["n_mel_channels"] = 40
    cfg["train"]["segment_size"] = 4096  # 8 frames
    if bucket:
        cfg["train"]["length_bucket_frames"] = bucket
    return cfg


def _ragged_batch(cfg, seed, gpu, B=4):
    """Collate-shaped batch whose padded lengths are the batch's OWN maxima (as the reference collate leaves them)."""
    from vcvits_amd import synthetic
    g = torch.Generator().manual_seed(seed)
    hop = cfg["data"]["hop_length"]
    ty = [int(torch.randint(40, 62, (1,), generator=g)) for _ in range(B)]  # spectrogram frames per utterance
    tx = [int(t * 0.53) for t in ty]                                          # content frames
    b = synthetic.full_batch(B, cfg["model"]["hubert_channels"], t_y=max(ty), t_x=max(tx), seed=seed)
    for i in range(B):
        b["y_wav_lengths"][i] = ty[i] * hop
        b["y_wav_values"][i, :, ty[i] * hop:] = 0.0
        b["x_hubert_features_lengths"][i] = b["x_pitch_lengths"][i] = tx[i]
        b["x_hubert_features_values"][i, :, tx[i]:] = 0.0
        b["x_pitch_values"][i, tx[i]:] = 0
    b["noise"] = torch.randn(B, cfg["model"]["inter_channels"], max(ty), generator=g)
    b["ids_slice"] = torch.tensor([int(torch.randint(0, t - 8, (1,), generator=g)) for t in ty])
    return {k: v.to(gpu) for k, v in b.items()}


def _run(cfg, sd, batches, gpu, graphs):
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VCVITS
    graphed.set_batch_enabled(graphs)
    torch.manual_seed(12)
    mod = VCVITS(**cfg)
    mod.load_state_dict(sd)
    mod = mod.to(gpu)
    mod.configure_optimizers()
    losses = []
    for b in batches:
        out = mod.fit_batch(b)
        losses.append((float(out["g"]), float(out["d"])))
    bg = mod.__dict__.get("_batch_graph")
    stats = (bg.replays, bg.captures, len(bg.entries), bg.failed) if bg is not None else (0, 0, 0, False)
    flat = (mod.optim_g.flat.clone(), mod.optim_d.flat.clone())
    mod.optim_g.close()
    mod.optim_d.close()
    return losses, stats, flat


@in_child
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_variable_length_stream_replays_and_matches_unbucketed_eager(gpu, dtype):
    from vcvits_amd import ops
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VCVITS
    torch.manual_seed(11)
    sd = copy.deepcopy(VCVITS(**_cfg(0)).state_dict())
    cfg_plain, cfg_bucket = _cfg(0), _cfg(32)
    batches = [_ragged_batch(cfg_plain, 300 + i, gpu) for i in range(14)]
    raw_shapes = {tuple(b["y_wav_values"].shape) + tuple(b["x_pitch_values"].shape) for b in batches}
    assert len(raw_shapes) >= 8  # the un-bucketed stream: (nearly) every batch its own shape
    ops.set_compute_dtype(dtype)
    was = graphed.BATCH_ENABLED[0]
    try:
        eager, st_e, flat_e = _run(cfg_plain, sd, batches, gpu, graphs=False)
        buck, st_b, flat_b = _run(cfg_bucket, sd, batches, gpu, graphs=True)
    finally:
        graphed.set_batch_enabled(was)
        ops.set_compute_dtype("f32")
    replays, captures, entries, failed = st_b
    assert st_e[0] == 0
    # 40..61 frames -> buckets of 32: padded to 64 (and content frames to 32 or 64): at most a couple of shapes, each recorded
    # on its third sighting and replayed from then on
    assert not failed and captures <= 3 and replays >= 6, st_b
    tol = 2e-3 if dtype == "f32" else 3e-2
    for i, ((ge, de), (gb, db)) in enumerate(zip(eager, buck)):
        assert abs(gb - ge) <= tol * abs(ge) + 1e-5 and abs(db - de) <= tol * abs(de) + 1e-5, (i, ge, gb, de, db)
    if dtype == "f32":
        # the first batch from identical state: the bucket padding alone (no optimizer history) -- fp32 rounding only
        assert abs(buck[0][0] - eager[0][0]) <= 2e-5 * abs(eager[0][0]) and abs(buck[0][1] - eager[0][1]) <= 2e-5 * abs(eager[0][1])
        for a, b in zip(flat_e, flat_b):
            # 14 AdamW steps of lr 2e-4 each: a parameter whose gradient sign is rounding noise can differ by 2 * 14 * lr
            assert (a - b).abs().max().item() <= 28 * 2e-4 + 1e-6
            assert ((a - b).abs() > 1e-4).float().mean().item() < 0.02
