"""One bench step under torch.profiler: torch-issued GPU ops by name and input shape, with the python call site."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from vcvits_amd import configs, ops, synthetic
from vcvits_amd.light.vcvits import VocoderGAN
dev = torch.device("cuda:0")
cfg = configs.base()
torch.manual_seed(1234)
module = VocoderGAN(**cfg).to(dev); module.train(); module.configure_optimizers()
batches = [synthetic.vocoder_batch(16, cfg["model"]["inter_channels"], seed=1234 + i, device=dev) for i in range(2)]
for i in range(3): module.fit_batch(batches[i % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    module.fit_batch(batches[1])
    torch.cuda.synchronize()

from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name in ("aten::add_", "aten::copy_", "aten::add", "aten::zero_", "aten::fill_", "aten::cat", "aten::mul", "aten::sum", "aten::clone", "aten::contiguous", "aten::zeros", "aten::empty_like", "aten::sub", "aten::neg", "aten::div", "aten::mean"):
        st = [f for f in (ev.stack or []) if "vcvits_amd" in f or "bench" in f][:3]
        key = (ev.name, str(ev.input_shapes)[:60], " <- ".join(x.split("/")[-1] for x in st))
        agg[key][0] += 1
        agg[key][1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:50]:
    print("%-12s %4d calls %8.1f us  %-60s %s" % (k[0], n, t, k[1], k[2]))
