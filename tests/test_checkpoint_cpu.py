"""CPU: checkpoint container / discovery / tolerant load."""
import os

import torch


def small_module():
    from vcvits_amd import configs
    from vcvits_amd.light.vcvits import VocoderGAN
    cfg = configs.base()
    cfg["model"].update({"inter_channels": 8, "upsample_initial_channel": 16, "multi_period_discriminator_periods": [2]})
    return VocoderGAN(**cfg)


def test_roundtrip_and_discovery(tmp_path):
    from vcvits_amd.light import checkpoint as ck
    torch.manual_seed(0)
    m = small_module()
    m.current_epoch, m.global_step = 3, 1234
    assert ck.last_checkpoint(str(tmp_path)) is None
    d0 = ck.next_version_dir(str(tmp_path))
    ck.save_checkpoint(m, os.path.join(d0, "last.ckpt"))
    d1 = ck.next_version_dir(str(tmp_path))
    assert d1.endswith(os.path.join("version_1", "checkpoints"))
    p1 = ck.save_checkpoint(m, os.path.join(d1, "last.ckpt"))
    assert ck.last_checkpoint(str(tmp_path)) == p1  # highest version wins (train.py:39-48)
    ck.save_checkpoint(m, os.path.join(d1, "epoch=2-step=10.ckpt"))
    assert ck.newest_ckpt_in(d1).endswith("last.ckpt")  # lexicographic (infer.py:13-14)
    raw = torch.load(p1, weights_only=False)
    assert set(raw) >= {"state_dict", "hyper_parameters", "epoch", "global_step", "optimizer_states"}
    assert "net_g.ups.0.weight_g" in raw["state_dict"] and "net_period_d.discriminators.1.convs.0.weight_v" in raw["state_dict"]
    assert raw["hyper_parameters"]["train"]["segment_size"] == 16384
    torch.manual_seed(1)
    m2 = small_module()
    assert not torch.equal(m2.net_g.conv_pre.weight, m.net_g.conv_pre.weight)
    ck.load_checkpoint(m2, p1)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert m2.current_epoch == 3 and m2.global_step == 1234


def test_tolerant_load_drops_mismatches(tmp_path):
    from vcvits_amd.light import checkpoint as ck
    m = small_module()
    p = ck.save_checkpoint(m, str(tmp_path / "a.ckpt"))
    raw = torch.load(p, weights_only=False)
    raw["state_dict"]["net_g.conv_pre.weight"] = torch.zeros(3, 3, 3)   # wrong shape -> keep fresh tensor
    raw["state_dict"]["net_g.not_a_parameter"] = torch.zeros(1)         # unknown key -> dropped
    raw["optimizer_states"] = [{"dummy": 1}]
    torch.save(raw, p)
    m2 = small_module()
    before = m2.net_g.conv_pre.weight.detach().clone()
    out = ck.load_checkpoint(m2, p)
    assert torch.equal(m2.net_g.conv_pre.weight, before)
    assert "optimizer_states" not in out
