"""Probe: CPU oracle vocoder batch (B=1) at a given torch thread count (bounded by the caller's timeout)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
n = int(sys.argv[1])
torch.set_num_threads(n)
from vcvits_amd import configs, synthetic
from vcvits_amd.light.vcvits import VocoderGAN, DEFAULT_PERIODS
from oracle.cpu_step import CpuTrainer
cfg = configs.base()
torch.manual_seed(0)
m = VocoderGAN(**cfg)
tr = CpuTrainer(m.state_dict(), cfg, DEFAULT_PERIODS, True)
b = synthetic.vocoder_batch(1, 256, seed=99)
t0 = time.perf_counter(); tr.batch(b); print("threads", n, "B=1 seconds", time.perf_counter() - t0, flush=True)
