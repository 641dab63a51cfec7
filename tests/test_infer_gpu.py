"""GPU: SynthesizerSVC.infer / voice_conversion (synthesizer_svc.py:90-119 of the reference) against
the oracle composition (content encoder -> prior sample -> flow reverse -> decoder), reduced widths."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import fill_state_dict, keys_shapes_of

pytestmark = pytest.mark.gpu


def close(name, a, b, tol=1e-4):
    a = a.detach().double().cpu()
    b = b.detach().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= tol * b.abs().max().item() + 2e-6, "%s: %.3e" % (name, err)


def test_infer_and_voice_conversion(gpu):
    from oracle import vits_oracle as O
    from vcvits_amd.model.synthesizers.synthesizer_svc import SynthesizerSVC
    C, H, HUB = 16, 16, 24
    net = SynthesizerSVC(1025, 8, C, H, 32, 2, 2, 3, 0.1, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 4, 2], 32,
                         [16, 16, 4, 4], HUB, 64, n_speakers=4, gin_channels=8, hubert_ckpt=None)
    sd = fill_state_dict(keys_shapes_of(net), 31)
    net.load_state_dict(sd)
    net = net.to(gpu).eval()
    rng = np.random.default_rng(3)
    T_x = 20
    feats = torch.from_numpy(rng.standard_normal((2, HUB, T_x)).astype(np.float32))
    x_len = torch.tensor([20, 15])
    pitch = torch.from_numpy(rng.integers(1, 64, size=(2, T_x)))
    sid = torch.tensor([1, 3])
    length_scale = 1.5
    y_len = (x_len * length_scale).long()
    T_y = int(y_len.max())
    noise = torch.from_numpy(rng.standard_normal((2, C, T_y)).astype(np.float32))
    with torch.no_grad():
        o, y_mask, (z, z_p, m_p, logs_p) = net.infer(feats.to(gpu), x_len.to(gpu), pitch.to(gpu), x_len.to(gpu),
                                                     sid=sid.to(gpu), length_scale=length_scale, max_len=25,
                                                     noise=noise.to(gpu))
        # oracle composition
        x, m, logs, x_mask = O.content_encoder_forward(sd_prefixed(sd, "n"), "n.enc_p", feats, x_len, pitch, C, 2, 2, 3)
        g = F.embedding(sid, sd["emb_g.weight"]).unsqueeze(-1)
        ym = O.sequence_mask(y_len, None).unsqueeze(1).float()
        m = F.interpolate(m, size=(T_y,), mode="nearest")
        logs = F.interpolate(logs, size=(T_y,), mode="nearest")
        zp_ref = m + noise * torch.exp(logs) * 1
        z_ref = O.flow_forward(sd_prefixed(sd, "n"), "n.flow", zp_ref, ym, g, True, C, H, 5, 1, 4)
        o_ref = O.generator_forward(sd_prefixed(sd, "n"), "n.dec", (z_ref * ym)[:, :, :25])
    close("m_p", m_p, m); close("z_p", z_p, zp_ref); close("z", z, z_ref)
    close("o", o, o_ref, tol=2e-4)
    assert o.shape == (2, 1, 25 * 512)
    # voice conversion: enc_q -> flow (src) -> flow reverse (tgt) -> dec
    spec = torch.from_numpy(np.abs(rng.standard_normal((2, 1025, 12))).astype(np.float32))
    spec_len = torch.tensor([12, 9])
    torch.manual_seed(5)
    with torch.no_grad():
        o_vc, mask_vc, (zq, zpq, zhat) = net.voice_conversion(spec.to(gpu), spec_len.to(gpu), sid.to(gpu),
                                                              torch.tensor([0, 2]).to(gpu))
        gm = O.sequence_mask(spec_len, 12).unsqueeze(1).float()
        g_src = F.embedding(sid, sd["emb_g.weight"]).unsqueeze(-1)
        g_tgt = F.embedding(torch.tensor([0, 2]), sd["emb_g.weight"]).unsqueeze(-1)
        zp_o = O.flow_forward(sd_prefixed(sd, "n"), "n.flow", zq.cpu(), gm, g_src, False, C, H, 5, 1, 4)
        zh_o = O.flow_forward(sd_prefixed(sd, "n"), "n.flow", zp_o, gm, g_tgt, True, C, H, 5, 1, 4)
        o_o = O.generator_forward(sd_prefixed(sd, "n"), "n.dec", zh_o * gm)
    close("vc z_p", zpq, zp_o); close("vc z_hat", zhat, zh_o); close("vc o", o_vc, o_o, tol=2e-4)


def sd_prefixed(sd, prefix):
    return {prefix + "." + k: v for k, v in sd.items()}
