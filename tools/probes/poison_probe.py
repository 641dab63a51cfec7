#!/usr/bin/env python3
"""GPU probe: does any result depend on the CONTENTS of uninitialised memory?

Every `torch.empty` / `torch.empty_like` / `torch.empty_strided` / `new_empty` result on the GPU is filled with NaN (floating
point) before it is handed out, then one eager training batch runs under autograd's anomaly detection: a kernel that reads
an element nobody wrote (a masked-out frame, a tile edge, the unspecified half of a restricted gradient) turns into a NaN in
the losses or the parameter gradients, and the anomaly trace names the backward node where it first appeared.  In eager
execution such reads see whatever an earlier tensor left behind -- finite, usually multiplied by a zero mask -- and go
unnoticed; in a HIP-graph replay they see the previous replay's leftovers at exactly the same addresses.

  python3 tools/probes/poison_probe.py --config 48k --workload full --dtype f32 --batch 16"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


class Poison:
    """Context: GPU tensors from torch.empty* come back filled with NaN (float) / a large negative value (integers)."""

    NAMES = ("empty", "empty_like", "empty_strided")

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in self.NAMES}
        self.saved_new_empty = torch.Tensor.new_empty

        def wrap(fn):
            def poisoned(*a, **k):
                t = fn(*a, **k)
                if t.is_cuda and t.numel() > 0:
                    if t.is_floating_point():
                        t.fill_(float("nan"))
                    elif t.dtype in (torch.int32, torch.int64):
                        t.fill_(-(1 << 30))
                return t
            return poisoned
        for n in self.NAMES:
            setattr(torch, n, wrap(self.saved[n]))
        torch.Tensor.new_empty = wrap(self.saved_new_empty)
        return self

    def __exit__(self, *exc):
        for n in self.NAMES:
            setattr(torch, n, self.saved[n])
        torch.Tensor.new_empty = self.saved_new_empty
        return False


def build(a, cfg, dev):
    from vcvits_amd.light.vcvits import VCVITS, VocoderGAN
    torch.manual_seed(1234)
    mod = (VocoderGAN if a.workload == "vocoder" else VCVITS)(**cfg)
    for mm in mod.modules():
        if hasattr(mm, "p_dropout"):
            mm.p_dropout = 0.0
    mod = mod.to(dev)
    mod.train()
    mod.configure_optimizers()
    return mod


def one_batch(mod, batch):
    grads = {}
    names = {id(p): n for n, p in mod.named_parameters()}

    def probe(idx, opt):
        for p in opt.params:
            grads[names[id(p)]] = p.grad.detach().clone()
    out = mod.fit_batch(batch, after_backward=probe)
    torch.cuda.synchronize()
    return {k: float(v) for k, v in out.items()}, grads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["base", "48k"], default="48k")
    ap.add_argument("--workload", choices=["vocoder", "full"], default="full")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--no-anomaly", action="store_true")
    a = ap.parse_args()
    from vcvits_amd import configs, ops, synthetic
    from vcvits_amd.light import graphed
    dev = torch.device("cuda:0")
    graphed.set_enabled(False)
    cfg = configs.base() if a.config == "base" else configs.base_48k()
    m = cfg["model"]
    ops.set_compute_dtype(a.dtype)
    make = synthetic.vocoder_batch if a.workload == "vocoder" else synthetic.full_batch
    width = m["inter_channels"] if a.workload == "vocoder" else m["hubert_channels"]
    batch = make(a.batch, width, seed=1234, device=dev)
    if a.workload == "full":
        g = torch.Generator().manual_seed(77)
        batch["noise"] = torch.randn(a.batch, m["inter_channels"], 384, generator=g).to(dev)
        batch["ids_slice"] = torch.randint(0, 250, (a.batch,), generator=g).to(dev)
    mod = build(a, cfg, dev)
    ref_loss, ref_grads = one_batch(mod, batch)
    mod.optim_g.close()
    mod.optim_d.close()
    del mod
    ops.invalidate_weights()
    mod = build(a, cfg, dev)
    print("config %s / %s / %s / B=%d: reference losses %s" % (a.config, a.workload, a.dtype, a.batch, ref_loss))
    try:
        with Poison():
            if a.no_anomaly:
                loss, grads = one_batch(mod, batch)
            else:
                with torch.autograd.detect_anomaly(check_nan=True):
                    loss, grads = one_batch(mod, batch)
    except RuntimeError as e:
        print("ANOMALY:", str(e)[:1500])
        return 1
    print("poisoned-allocation losses %s" % loss)
    bad = []
    worst = 0.0
    for n, gref in ref_grads.items():
        gp = grads[n]
        if not bool(torch.isfinite(gp).all()):
            bad.append((n, int((~torch.isfinite(gp)).sum()), gp.numel()))
            continue
        worst = max(worst, float((gp - gref).abs().max()) / (float(gref.abs().max()) + 1e-30))
    for n, k, tot in bad[:40]:
        print("  NON-FINITE gradient: %-60s %d of %d elements" % (n, k, tot))
    print("%d of %d parameter gradients non-finite; worst relative difference of the finite ones %.2e" % (len(bad), len(ref_grads), worst))
    return 1 if bad or not all(v == v for v in loss.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
