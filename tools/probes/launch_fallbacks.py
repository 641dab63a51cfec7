"""Per-step log of the conv launches that (a) miss the pack cache or (b) no packed-weight kernel family takes (they run on the\ngeneric register-staged kernel): python tools/probes/launch_fallbacks.py   (vocoder workload, base config, B = 16)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vcvits_amd import configs, ops, synthetic
from vcvits_amd.light.vcvits import VocoderGAN
dev = torch.device("cuda:0")
cfg = configs.base()
torch.manual_seed(1234)
module = VocoderGAN(**cfg).to(dev)
module.train(); module.configure_optimizers()
m = cfg["model"]
batches = [synthetic.vocoder_batch(16, m["inter_channels"], seed=1234 + i, device=dev) for i in range(2)]
for i in range(4):
    module.fit_batch(batches[i % 2])
torch.cuda.synchronize()
stat = collections.Counter()
orig_entry = ops._stable_entry
cur = {}
def entry(w):
    e = orig_entry(w)
    cur["e"] = e
    return e
ops.replace("_stable_entry", entry)
orig_launch = ops._launch_conv
def launch(a, flip_w=None, wt=None):
    cur["e"] = "unset"
    before = dict(ops.LAUNCH_COUNTS)
    n_packs = {id(e): len(e["packs"]) for e in ops._WN_CACHE.values()}
    orig_launch(a, flip_w, wt)
    e = cur.get("e")
    fam = [k for k in ops.LAUNCH_COUNTS if ops.LAUNCH_COUNTS[k] != before.get(k, 0)]
    if e == "unset":
        stat[("no packed family", tuple(fam), a.Cg, a.Mg, a.K, a.B, a.Tout, a.P)] += 1
    elif e is None:
        stat[("NO ENTRY (plain weight)", tuple(fam), a.Cg, a.Mg, a.K, a.B, a.Tout, a.P, flip_w is not None)] += 1
    elif len(e["packs"]) != n_packs.get(id(e)):
        stat[("MISS in entry", tuple(fam), a.Cg, a.Mg, a.K, a.B, a.Tout, a.P, flip_w is not None)] += 1
ops.replace("_launch_conv", launch)
for i in range(2):
    module.fit_batch(batches[i % 2])
torch.cuda.synchronize()
for k, v in sorted(stat.items(), key=lambda kv: -kv[1]):
    print(v / 2.0, k)
