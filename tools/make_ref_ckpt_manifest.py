#!/usr/bin/env python3
"""Writes tests/golden/ref_ckpt_manifest.json: the state_dict key order, shapes and parameter/buffer kinds of the
REFERENCE module tree (`VCVITS` of /root/reference/vits/light/vcvits.py), from which tests/test_checkpoint_cpu.py
synthesises a Lightning-shaped `.ckpt` dict (torch.optim `optimizer_states` indexed in this order) without needing
the reference at test time.  A manifest is names and shapes -- data, not source.

Built here from imports of the reference's own classes (relative_attention_transformer.TransformerEncoder,
PosteriorEncoder, ResidualCouplingBlock, MultiPeriodDiscriminator, MultiScaleDiscriminator) assembled in the
registration order of the constructors that cannot be imported offline (content_encoder.py:14-50 needs fairseq,
synthesizer_svc.py:19-67 needs torchaudio + a torch.hub download, vcvits.py:28-52 needs Lightning):
  VCVITS:          net_g, net_period_d, net_scale_d, audio_pipeline                     (vcvits.py:33-52)
  SynthesizerSVC:  enc_p, dec, enc_q, flow, emb_g                                       (synthesizer_svc.py:57-67)
  HubertContentEncoder: hubert, hubert_proj, emb_pitch, encoder, proj                   (content_encoder.py:32-50)
  audio_pipeline:  spec.window, invspec.window (torchaudio Spectrogram / InverseSpectrogram buffers, pipeline.py:23-27)
Stand-ins, marked "third_party" in the manifest: `hubert` is a 3-tensor stub (the fairseq model is not available; its
entries only have to be skipped and counted), `dec` uses this build's Generator key names (canonical HiFi-GAN names;
the hub model's own names cannot be checked offline).

Run in the authoring container: python tools/make_ref_ckpt_manifest.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
import torch  # noqa: E402
from torch import nn  # noqa: E402

from vits.model.discriminators.multi_period_discriminator import MultiPeriodDiscriminator  # noqa: E402
from vits.model.discriminators.multi_scale_discriminator import MultiScaleDiscriminator  # noqa: E402
from vits.model.encoders.posterior_encoder import PosteriorEncoder  # noqa: E402
from vits.model.flow import ResidualCouplingBlock  # noqa: E402
from vits.model.transformer.relative_attention_transformer import TransformerEncoder  # noqa: E402

# reduced widths for net_g (constructor arguments are free); the discriminators' widths are fixed by the reference
C, H, FILT, HEADS, LAYERS, HUB, NPITCH, NSPK, GIN, UPC = 16, 16, 32, 2, 2, 24, 64, 8, 8, 32
PERIODS = [2]


class HubertStub(nn.Module):
    def __init__(self):
        super().__init__()
        self.mask_emb = nn.Parameter(torch.zeros(HUB))
        self.feature_extractor = nn.Sequential(nn.Conv1d(1, 4, 10, 5, bias=False))
        self.final_proj = nn.Linear(HUB, 4)


class EncP(nn.Module):  # registration order of content_encoder.py:32-50
    def __init__(self):
        super().__init__()
        self.hubert = HubertStub()
        self.hubert_proj = nn.Linear(HUB, H)
        self.emb_pitch = nn.Embedding(NPITCH, H)
        self.encoder = TransformerEncoder(H, FILT, HEADS, LAYERS, 3, 0.1)
        self.proj = nn.Conv1d(H, C * 2, 1)


def product_generator():
    sys.path.insert(0, ROOT)
    sys.modules.pop("vits", None)
    from vcvits_amd.model.generator import Generator
    return Generator(C, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 4, 2], UPC, [16, 16, 4, 4])


class NetG(nn.Module):  # synthesizer_svc.py:57-67
    def __init__(self):
        super().__init__()
        self.enc_p = EncP()
        self.dec = nn.Module()  # placeholder, replaced below (product import must come after the reference imports)
        self.enc_q = PosteriorEncoder(1025, C, H, 5, 1, 16, gin_channels=GIN)
        self.flow = ResidualCouplingBlock(C, H, 5, 1, 4, gin_channels=GIN)
        self.emb_g = nn.Embedding(NSPK, GIN)


class Pipe(nn.Module):
    def __init__(self):
        super().__init__()
        self.spec, self.invspec = nn.Module(), nn.Module()
        self.spec.register_buffer("window", torch.hann_window(2048))
        self.invspec.register_buffer("window", torch.hann_window(2048))


class Tree(nn.Module):  # vcvits.py:33-52
    def __init__(self):
        super().__init__()
        self.net_g = NetG()
        self.net_period_d = MultiPeriodDiscriminator(periods=PERIODS, use_spectral_norm=False)
        self.net_scale_d = MultiScaleDiscriminator(False)
        self.audio_pipeline = Pipe()


tree = Tree()
dec = product_generator()
# keep `dec` in its registration slot (between enc_p and enc_q)
tree.net_g._modules["dec"] = dec
params = {n for n, _ in tree.named_parameters()}
entries = []
for k, v in tree.state_dict().items():
    third = k.startswith("net_g.enc_p.hubert.") or k.startswith("audio_pipeline.") or k.startswith("net_g.dec.")
    entries.append({"key": k, "shape": list(v.shape), "param": k in params, "third_party": third})
order = {"net_g": [n for n, _ in tree.net_g.named_parameters()],
         "disc": [n for n, _ in tree.net_period_d.named_parameters()] + [n for n, _ in tree.net_scale_d.named_parameters()]}
out = {"widths": dict(C=C, H=H, FILT=FILT, HEADS=HEADS, LAYERS=LAYERS, HUB=HUB, NPITCH=NPITCH, NSPK=NSPK, GIN=GIN,
                      UPC=UPC, PERIODS=PERIODS),
       "entries": entries, "n_params_g": len(order["net_g"]), "n_params_d": len(order["disc"])}
path = os.path.join(ROOT, "tests", "golden", "ref_ckpt_manifest.json")
json.dump(out, open(path, "w"), indent=0)
print(path, len(entries), "entries;", out["n_params_g"], "G params,", out["n_params_d"], "D params")
