"""GPU: an embedding index outside its table.  The reference's nn.Embedding (content_encoder.py:40 emb_pitch,
synthesizer_svc.py:68 emb_g) raises on one; the HIP lookup zero-fills the column, counts the position in a device word and
the host raises at its next check point (ops.check_indices: validation, checkpoint save, epoch end; every batch with
VCVITS_CHECK_INDICES=1)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_out_of_range_index_is_counted_and_raised(gpu):
    from vcvits_amd import ops
    ops.index_errors()  # (clear what earlier tests may have left)
    W = torch.randn(16, 8, device=gpu)
    idx = torch.tensor([[0, 3, 15, 7], [1, 2, 4, 5]], device=gpu)
    y = ops.embedding_t(idx, W)
    assert torch.equal(y, W[idx].transpose(1, 2))
    assert ops.index_errors() == 0
    ops.check_indices()
    bad = idx.clone()
    bad[0, 1], bad[1, 3] = 16, -1
    y = ops.embedding_t(bad, W)
    ref = W[bad.clamp(0, 15)].transpose(1, 2).clone()
    ref[0, :, 1] = 0
    ref[1, :, 3] = 0
    assert torch.equal(y, ref)  # (the zero-filled columns)
    with pytest.raises(IndexError, match="2 embedding lookup"):
        ops.check_indices()
    ops.check_indices()  # the count was reset by the raise


def test_checkpoint_save_refuses_after_a_bad_speaker_id(gpu, tmp_path):
    from vcvits_amd import configs, ops, synthetic
    from vcvits_amd.light import checkpoint
    from vcvits_amd.light.vcvits import VCVITS
    ops.index_errors()
    c = configs.base()
    c["model"].update({"inter_channels": 16, "hidden_channels": 16, "filter_channels": 32, "n_heads": 2, "n_layers": 1,
                       "upsample_initial_channel": 32, "hubert_channels": 24, "gin_channels": 8, "p_dropout": 0.0,
                       "multi_period_discriminator_periods": [2]})
    c["data"].update({"n_mel_channels": 40, "hubert_channels": 24, "n_speakers": 8})
    c["train"]["segment_size"] = 4096
    torch.manual_seed(0)
    m = VCVITS(**c).to(gpu)
    m.configure_optimizers()
    batch = synthetic.full_batch(2, 24, t_y=64, t_x=40, device=gpu)
    batch["sid"] = torch.tensor([3, 8], device=gpu)  # 8 == n_speakers: one past the table
    m.fit_batch(batch)
    with pytest.raises(IndexError):
        checkpoint.save_checkpoint(m, str(tmp_path / "last.ckpt"))
    batch["sid"] = torch.tensor([3, 7], device=gpu)
    m.fit_batch(batch)
    checkpoint.save_checkpoint(m, str(tmp_path / "last.ckpt"))
