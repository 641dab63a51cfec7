"""Race detector for the multi-stream discriminators inside a recorded batch: in deterministic mode (no atomics anywhere in
the GAN step) the replayed batches must equal the eager loop BIT FOR BIT, whatever the stream count.
  VCVITS_STREAMS=2 python tools/probes/streams_race_probe.py [--full] [--one-batch]"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from vcvits_amd import configs, ops, synthetic  # noqa: E402
from vcvits_amd.light import graphed  # noqa: E402
from vcvits_amd.light.vcvits import VCVITS, VocoderGAN  # noqa: E402


def _force_recording():
    """GraphedBatch refuses to record with VCVITS_STREAMS > 1 (the replays diverge: this probe shows it); look the other way
    while it asks."""
    orig = graphed.GraphedBatch.applicable

    def applicable(self):
        old = os.environ.get("VCVITS_STREAMS")
        os.environ["VCVITS_STREAMS"] = "1"
        try:
            return orig(self)
        finally:
            if old is None:
                os.environ.pop("VCVITS_STREAMS", None)
            else:
                os.environ["VCVITS_STREAMS"] = old
    graphed.GraphedBatch.applicable = applicable


def main():
    _force_recording()
    nb = 1 if "--one-batch" in sys.argv else 2
    full = "--full" in sys.argv
    dev = torch.device("cuda:0")
    ops.set_deterministic(True)
    cfg = configs.base_48k() if "--48k" in sys.argv else configs.base()
    if full:
        cfg["model"]["p_dropout"] = 0.0
    torch.manual_seed(3)
    cls = VCVITS if full else VocoderGAN
    sd = copy.deepcopy(cls(**cfg).state_dict())
    B = 16 if "--b16" in sys.argv else 4
    m = cfg["model"]
    if full:
        batches = []
        for i in range(2):
            b = synthetic.full_batch(B, m["hubert_channels"], seed=40 + i, device=dev)
            g = torch.Generator().manual_seed(70 + i)
            b["noise"] = torch.randn(B, m["inter_channels"], 384, generator=g).to(dev)
            b["ids_slice"] = torch.tensor(([5, 100, 17, 200] * 4)[:B], device=dev)
            batches.append(b)
    else:
        batches = [synthetic.vocoder_batch(B, m["inter_channels"], seed=40 + i, device=dev) for i in range(2)]
    res = {}
    for mode in (False, True):
        graphed.set_batch_enabled(mode)
        mod = cls(**cfg)
        mod.load_state_dict(sd)
        mod = mod.to(dev)
        mod.configure_optimizers()
        ls = []
        snap = None
        for i in range(10):
            o = mod.fit_batch(batches[i % nb])
            ls.append((o["g"].item(), o["d"].item()))
            if i == 2:  # the first replayed batch: the gradients it left in the flat buffers
                names = {id(p): n for n, p in mod.named_parameters()}
                snap = {}
                for tag, opt in (("G", mod.optim_g), ("D", mod.optim_d)):
                    for p, off in zip(opt.params, opt.offsets):
                        snap[tag + ":" + names[id(p)]] = opt.grad[off:off + p.numel()].clone()
        bg = mod.__dict__.get("_batch_graph")
        res[mode] = (ls, mod.optim_g.flat.clone(), mod.optim_d.flat.clone(), bg.replays if bg is not None else 0, snap)
        mod.optim_g.close()
        mod.optim_d.close()
    (l0, g0, d0, r0, s0), (l1, g1, d1, r1, s1) = res[False], res[True]
    bad = [(k, float((s0[k] - s1[k]).abs().max()), float(s0[k].abs().max())) for k in s0 if not torch.equal(s0[k], s1[k])]
    print("gradient tensors of the first replayed batch that differ from the eager batch's: %d of %d" % (len(bad), len(s0)))
    for k, e, sc in bad[:40]:
        print("   %-60s max |diff| %.3e  (scale %.3e)" % (k, e, sc))
    print("streams", os.environ.get("VCVITS_STREAMS", "1"), "full" if full else "vocoder", "replays", r1)
    same = all(a == b for a, b in zip(l0, l1))
    print("checksums of the final parameters (compare across stream counts): eager G %.10e D %.10e | replayed G %.10e D %.10e"
          % (float(g0.double().sum()), float(d0.double().sum()), float(g1.double().sum()), float(d1.double().sum())))
    print("losses bit-identical:", same, "| generator params max |diff|", float((g0 - g1).abs().max()), "| discriminator params max |diff|", float((d0 - d1).abs().max()))
    if not same:
        for i, (a, b) in enumerate(zip(l0, l1)):
            print(i, a, b)


if __name__ == "__main__":
    main()
