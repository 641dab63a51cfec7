"""Race detector for the multi-stream discriminators inside a recorded batch: in deterministic mode (no atomics anywhere in
the GAN step) the replayed batches must equal the eager loop BIT FOR BIT, whatever the stream count.
  VCVITS_STREAMS=2 python tools/probes/streams_race_probe.py [--full]"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from vcvits_amd import configs, ops, synthetic  # noqa: E402
from vcvits_amd.light import graphed  # noqa: E402
from vcvits_amd.light.vcvits import VCVITS, VocoderGAN  # noqa: E402


def main():
    full = "--full" in sys.argv
    dev = torch.device("cuda:0")
    ops.set_deterministic(True)
    cfg = configs.base()
    if full:
        cfg["model"]["p_dropout"] = 0.0
    torch.manual_seed(3)
    cls = VCVITS if full else VocoderGAN
    sd = copy.deepcopy(cls(**cfg).state_dict())
    B = 4
    m = cfg["model"]
    if full:
        batches = []
        for i in range(2):
            b = synthetic.full_batch(B, m["hubert_channels"], seed=40 + i, device=dev)
            g = torch.Generator().manual_seed(70 + i)
            b["noise"] = torch.randn(B, m["inter_channels"], 384, generator=g).to(dev)
            b["ids_slice"] = torch.tensor([5, 100, 17, 200], device=dev)
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
        for i in range(10):
            o = mod.fit_batch(batches[i % 2])
            ls.append((o["g"].item(), o["d"].item()))
        bg = mod.__dict__.get("_batch_graph")
        res[mode] = (ls, mod.optim_g.flat.clone(), mod.optim_d.flat.clone(), bg.replays if bg is not None else 0)
        mod.optim_g.close()
        mod.optim_d.close()
    (l0, g0, d0, r0), (l1, g1, d1, r1) = res[False], res[True]
    print("streams", os.environ.get("VCVITS_STREAMS", "1"), "full" if full else "vocoder", "replays", r1)
    same = all(a == b for a, b in zip(l0, l1))
    print("losses bit-identical:", same, "| generator params max |diff|", float((g0 - g1).abs().max()), "| discriminator params max |diff|", float((d0 - d1).abs().max()))
    if not same:
        for i, (a, b) in enumerate(zip(l0, l1)):
            print(i, a, b)


if __name__ == "__main__":
    main()
