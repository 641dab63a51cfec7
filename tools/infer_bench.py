"""BASELINE.json configs[4]: infer.py voice-conversion path -- flow inverse + HiFi-GAN decode at
48 kHz widths, batch x 10 s utterances, 1 x MI355X -> real-time factor.  fp32 (the bf16 build of
the kernels does not exist yet).  python tools/infer_bench.py [--batch 64] [--reps 3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vcvits_amd import configs
from vcvits_amd.model.synthesizers.synthesizer_svc import SynthesizerSVC

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--frames", type=int, default=938)  # 10 s at 48 kHz / hop 512
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = configs.base_48k()
torch.manual_seed(0)
d, m = cfg["data"], cfg["model"]
net = SynthesizerSVC(d["filter_length"] // 2 + 1, 32, n_speakers=d["n_speakers"], **m).to(dev).eval()
with torch.no_grad():
    for p in net.flow.parameters():
        if p.abs().sum() == 0:
            p.normal_(0, 0.02)
B, T = a.batch, a.frames
m_p = torch.randn(B, m["inter_channels"], T, device=dev)
logs_p = torch.randn(B, m["inter_channels"], T, device=dev) * 0.1 - 1.0
y_mask = torch.ones(B, 1, T, device=dev)
g = net.emb_g(torch.randint(0, 512, (B,), device=dev)).unsqueeze(-1)


def run():
    with torch.no_grad():
        z_p = m_p + torch.randn_like(m_p) * torch.exp(logs_p)
        z = net.flow(z_p, y_mask, g=g, reverse=True)
        return net.dec(z)


o = run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    o = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
audio_s = B * o.shape[-1] / d["target_sampling_rate"]
print(json.dumps({"metric": "inference real-time factor (flow inverse + HiFi-GAN decode, 48k_base widths)",
                  "value": round(dt / audio_s, 6), "unit": "wall s / audio s", "batch": B, "frames": T,
                  "samples_out": int(o.shape[-1]), "seconds_per_batch": round(dt, 4), "dtype": "f32",
                  "algorithmic_tflops": round(B * 770.0 / 1e3 / dt, 2), "higher_is_better": False}))
