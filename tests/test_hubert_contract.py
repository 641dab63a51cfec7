"""f1: the HuBERT feature contract of content_encoder.py:53-58 behind a pluggable extractor, and
the reference's batch schema (`x_wav_values`, vcvits.py:55-62) driving `training_step` unchanged.  The fairseq model
itself is third-party; a small frozen conv stack with its `extract_features` contract (receptive field 400, hop 320)
stands in for it on both sides."""
import copy

import pytest
import torch


class StubHubert(torch.nn.Module):
    """extract_features(wav [B, T]) -> (feats [B, T', H], padding_mask): window 400 / hop 320 like HuBERT's conv
    front end, so T' = (T - 400) // 320 + 1 = T_src // 320 after the 40 + 40 sample pad."""

    def __init__(self, channels):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.weight = torch.nn.Parameter(torch.randn(channels, 1, 400, generator=g) * 0.05, requires_grad=False)

    def extract_features(self, wav, padding_mask=None, mask=False, output_layer=None):
        f = torch.nn.functional.conv1d(wav.unsqueeze(1), self.weight, stride=320)
        return torch.tanh(f).transpose(1, 2), None


def test_extract_contract_cpu():
    """Pure host logic (no kernels): pad 40+40, extractor call on [B, T], transposes; errors are explicit."""
    from oracle import vits_oracle as O
    from vcvits_amd.model.encoders.content_encoder import HubertContentEncoder
    enc = HubertContentEncoder(None, 8, 8, 16, 2, 1, 3, 0.0, 24, 64)
    wav = torch.randn(2, 1, 3200)
    with pytest.raises(RuntimeError, match="feature extractor"):
        enc.extract(wav)
    stub = StubHubert(24)
    enc.set_feature_extractor(stub)
    feats = enc.extract(wav)
    assert feats.shape == (2, 24, 10) and feats.is_contiguous()
    assert torch.equal(feats, O.hubert_features(stub, wav))
    assert not any(k.startswith("hubert.") or "_extractor" in k for k in enc.state_dict())
    assert all("stub" not in n for n, _ in enc.named_parameters())
    enc.set_feature_extractor(lambda w: stub.extract_features(w)[0])  # a plain callable works too
    assert torch.equal(enc.extract(wav), feats)
    enc.set_feature_extractor(lambda w: torch.zeros(2, 10, 7))
    with pytest.raises(RuntimeError, match="expected"):
        enc.extract(wav)


def test_batch_schema_error():
    from vcvits_amd import configs
    from vcvits_amd.light.vcvits import VCVITS
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 8, "hidden_channels": 8, "filter_channels": 16, "n_heads": 2, "n_layers": 1,
                         "upsample_initial_channel": 16, "hubert_channels": 24, "gin_channels": 8,
                         "multi_period_discriminator_periods": [2]})
    m = VCVITS(**cfg)
    with pytest.raises(KeyError, match="x_wav_values"):
        m._source({"y_wav_values": torch.zeros(1)})


@pytest.mark.gpu
def test_training_step_from_reference_batch(gpu):
    """A batch in the reference's schema (collate.py:177-187) through audio_pipeline + extractor + the whole G/D
    step, against the CPU oracle fed with features the oracle's restatement of the same front end produced."""
    from oracle import vits_oracle as O
    from oracle.cpu_step import CpuTrainer
    from test_training_step_gpu import _run, small_cfg
    from vcvits_amd.light.vcvits import VCVITS
    torch.manual_seed(2)
    cfg = small_cfg()
    module = VCVITS(**cfg)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if ".post." in n:
                p.normal_(0.0, 0.05)
    trainer = CpuTrainer(copy.deepcopy(module.state_dict()), cfg, [2, 3], vocoder_only=False)
    stub = StubHubert(24)
    module = module.to(gpu)
    module.set_feature_extractor(copy.deepcopy(stub).to(gpu))
    module.configure_optimizers()
    g = torch.Generator().manual_seed(7)
    T_src = 22 * 320
    x_wav = torch.rand(2, 1, T_src, generator=g) * 1.6 - 0.8
    x_wav[1, :, 18 * 320:] = 0
    batch = {"sid": torch.tensor([1, 5]), "x_wav_values": x_wav, "x_wav_lengths": torch.tensor([T_src, 18 * 320]),
             "x_pitch_values": torch.randint(1, 512, (2, 22), generator=g), "x_pitch_lengths": torch.tensor([22, 18]),
             "y_wav_values": torch.rand(2, 1, 40 * 512, generator=g) * 1.8 - 0.9,
             "y_wav_lengths": torch.tensor([40 * 512, 30 * 512]),
             "noise": torch.randn(2, 16, 40, generator=g), "ids_slice": torch.tensor([3, 11])}
    batch["y_wav_values"][1, :, 30 * 512:] = 0
    feats = O.hubert_features(stub, O.audio_pipeline(x_wav))
    assert feats.shape == (2, 24, 22)
    ref_batch = {k: v for k, v in batch.items() if not k.startswith("x_wav")}
    ref_batch["x_hubert_features_values"] = feats
    ref_batch["x_hubert_features_lengths"] = batch["x_wav_lengths"]  # sample counts: the reference's quirk (D.8)

    class Both:  # the trainer sees the feature batch, the module the reference-schema batch
        def batch(self, _b):
            return trainer.batch(ref_batch)
        grads_g = property(lambda s: trainer.grads_g)
        grads_d = property(lambda s: trainer.grads_d)
    _run(module, Both(), batch, gpu)
