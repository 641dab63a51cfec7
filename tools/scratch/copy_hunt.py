"""Which python lines issue device copies in one training step (monkeypatched Tensor methods)."""
import os, sys, collections, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vcvits_amd import configs, synthetic
from vcvits_amd.light.vcvits import VocoderGAN
cfg = configs.base()
m = VocoderGAN(**cfg).to("cuda")
m.configure_optimizers()
mm = cfg["model"]
batches = [synthetic.vocoder_batch(16, mm["inter_channels"], seed=1 + i, device=torch.device("cuda")) for i in range(2)]
for i in range(2):
    m.fit_batch(batches[i % 2])
torch.cuda.synchronize()
cnt = collections.Counter()
def where():
    for fs in reversed(traceback.extract_stack()[:-2]):
        if "vcvits_amd" in fs.filename:
            return "%s:%d %s" % (os.path.relpath(fs.filename, ROOT), fs.lineno, fs.line)
    return "?"
def wrap(cls, name):
    orig = getattr(cls, name)
    def f(self, *a, **k):
        if isinstance(self, torch.Tensor) and self.is_cuda and not (name == "contiguous" and self.is_contiguous()):
            cnt[(name, where())] += 1
        return orig(self, *a, **k)
    setattr(cls, name, f)
for n in ("contiguous", "clone", "copy_", "to"):
    wrap(torch.Tensor, n)
oc = torch.cat
def cat(*a, **k):
    cnt[("cat", where())] += 1
    return oc(*a, **k)
torch.cat = cat
m.fit_batch(batches[0])
torch.cuda.synchronize()
for (n, s), c in cnt.most_common(30):
    print("%4d %-12s %s" % (c, n, s))
