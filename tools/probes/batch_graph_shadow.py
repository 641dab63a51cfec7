#!/usr/bin/env python3
"""GPU probe: a HIP-graph replay of the training batch against the eager batch FROM THE SAME STATE, parameter by parameter.

Two modules: A runs `fit_batch` with batch graphs on, B (the shadow) eagerly.  Before every batch B's whole optimizer
state (parameters, both moments, per-parameter step counts) is overwritten with A's, so the two batches start from
identical state; afterwards the gradient buffers, the updated parameters and the losses are compared.  The first replay
whose gradients differ from the eager batch's by more than summation-order noise names the parameters that went wrong.

  python3 tools/probes/batch_graph_shadow.py --config 48k --workload full --dtype f32 --batch 16 --steps 12"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["base", "48k"], default="base")
    ap.add_argument("--workload", choices=["vocoder", "full"], default="full")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--top", type=int, default=6)
    a = ap.parse_args()
    from vcvits_amd import configs, ops, synthetic
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VCVITS, VocoderGAN
    dev = torch.device("cuda:0")
    cfg = configs.base() if a.config == "base" else configs.base_48k()
    m = cfg["model"]
    ops.set_compute_dtype(a.dtype)
    make = synthetic.vocoder_batch if a.workload == "vocoder" else synthetic.full_batch
    width = m["inter_channels"] if a.workload == "vocoder" else m["hubert_channels"]
    batches = [make(a.batch, width, seed=1234 + i, device=dev) for i in range(2)]
    if a.workload == "full":
        for i, b in enumerate(batches):
            g = torch.Generator().manual_seed(77 + i)
            b["noise"] = torch.randn(a.batch, m["inter_channels"], 384, generator=g).to(dev)
            b["ids_slice"] = torch.randint(0, 250, (a.batch,), generator=g).to(dev)
    mods = []
    for _ in range(2):
        torch.manual_seed(1234)
        mod = (VocoderGAN if a.workload == "vocoder" else VCVITS)(**cfg)
        for mm in mod.modules():
            if hasattr(mm, "p_dropout"):
                mm.p_dropout = 0.0
        mod = mod.to(dev)
        mod.train()
        mod.configure_optimizers()
        mods.append(mod)
    A, B = mods
    names = {}
    for tag, opt_name in (("g", "optim_g"), ("d", "optim_d")):
        opt = getattr(A, opt_name)
        byid = {id(p): n for n, p in A.named_parameters()}
        names[tag] = [(byid[id(p)], off, p.numel()) for p, off in zip(opt.params, opt.offsets)]
    print("config %s / %s / %s / B=%d" % (a.config, a.workload, a.dtype, a.batch))
    for i in range(a.steps):
        for tag in ("optim_g", "optim_d"):
            oa, ob = getattr(A, tag), getattr(B, tag)
            ob.flat.copy_(oa.flat)
            ob.exp_avg.copy_(oa.exp_avg)
            ob.exp_avg_sq.copy_(oa.exp_avg_sq)
            ob._pstep = list(oa._pstep)
            ob.step_count = oa.step_count
        ops.invalidate_weights()
        torch.cuda.synchronize()
        bg = A.__dict__.get("_batch_graph")
        r0 = bg.replays if bg is not None else 0
        graphed.set_enabled(True)
        graphed.set_batch_enabled(True)
        outA = {k: v.clone() for k, v in A.fit_batch(batches[i % 2]).items()}
        bg = A.__dict__.get("_batch_graph")
        replayed = bg is not None and bg.replays > r0
        torch.cuda.synchronize()
        graphed.set_enabled(False)
        outB = {k: v.clone() for k, v in B.fit_batch(batches[i % 2]).items()}
        torch.cuda.synchronize()
        graphed.set_enabled(True)
        line = "step %2d %s  loss g %.6f / %.6f  d %.6f / %.6f" % (i, "REPLAY" if replayed else "eager ", float(outA["g"]), float(outB["g"]),
                                                                 float(outA["d"]), float(outB["d"]))
        worst = []
        for tag, opt_name in (("g", "optim_g"), ("d", "optim_d")):
            oa, ob = getattr(A, opt_name), getattr(B, opt_name)
            for what, ta, tb in (("grad", oa.grad, ob.grad), ("param", oa.flat, ob.flat)):
                d = (ta - tb).abs()
                line += "  |%s %s| max diff %.2e (max %.2e)" % (tag, what, float(d.max()), float(tb.abs().max()))
                if what == "grad":
                    for n, off, k in names[tag]:
                        sa, sb = ta[off:off + k], tb[off:off + k]
                        den = float(sb.abs().max()) + 1e-30
                        e = float((sa - sb).abs().max()) / den
                        if not (e < 1e30):
                            e = float("inf")
                        worst.append((e, tag + ":" + n, den))
        print(line)
        worst.sort(key=lambda t: -t[0])
        for e, n, den in worst[:a.top]:
            print("      grad %-60s rel max err %.2e (scale %.2e)" % (n, e, den))
    for mod in mods:
        mod.optim_g.close()
        mod.optim_d.close()


if __name__ == "__main__":
    main()
