#!/usr/bin/env python3
"""GPU probe: the whole-batch HIP graph (vcvits_amd/light/graphed.py: GraphedBatch) at benchmark widths.

  python3 tools/probes/batch_graph_probe.py --config 48k --workload full --dtype bf16 --batch 16 --steps 16

Runs the same `steps` batches from the same initial state twice -- all eager, then with batch graphs on and every
`--eager-every`-th batch forced eager (the pattern bench.py's sampled profiler steps produce, the one round 4's step
graphs faulted in) -- and prints the losses side by side, the host time to issue a batch in each mode and the device time
per batch.  VCVITS_CHECK_PTRS=1 lists every tensor from outside the graph's pool that the recorded launches bake."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["base", "48k"], default="base")
    ap.add_argument("--workload", choices=["vocoder", "full"], default="vocoder")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--eager-every", type=int, default=4)
    ap.add_argument("--second", choices=["graph", "eager"], default="graph",
                    help="eager: compare the eager loop with ITSELF (how far two runs drift apart on their own: atomics)")
    ap.add_argument("--ddp", action="store_true", help="1-rank RCCL group with the bucket hooks on (VCVITS_FORCE_DDP=1)")
    a = ap.parse_args()
    if a.ddp:
        os.environ["VCVITS_FORCE_DDP"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
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
        for i, b in enumerate(batches):  # the two random draws injected: both loops see the same ones
            g = torch.Generator().manual_seed(77 + i)
            b["noise"] = torch.randn(a.batch, m["inter_channels"], 384, generator=g).to(dev)
            b["ids_slice"] = torch.randint(0, 250, (a.batch,), generator=g).to(dev)
    res = {}
    for mode in ("eager", "graph"):
        if mode == "graph" and a.second == "eager":
            graphed.set_enabled(False)
        torch.manual_seed(1234)
        mod = (VocoderGAN if a.workload == "vocoder" else VCVITS)(**cfg)
        if a.workload == "full":
            for mm in mod.modules():  # dropout off: the two loops draw masks from different streams otherwise
                if hasattr(mm, "p_dropout"):
                    mm.p_dropout = 0.0
        mod = mod.to(dev)
        mod.train()
        mod.configure_optimizers()
        losses, issue = [], []
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for i in range(a.steps):
            graphed.set_batch_enabled(mode == "graph" and (a.eager_every <= 0 or i % a.eager_every != a.eager_every - 1))
            t0 = time.perf_counter()
            out = mod.fit_batch(batches[i % 2])
            issue.append(time.perf_counter() - t0)
            losses.append({k: v.clone() for k, v in out.items()})  # (a replay's losses are static tensors)
            if i == a.steps // 2:
                torch.cuda.synchronize()
                t_half = time.perf_counter()
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        losses = [(float(o["g"]), float(o["d"])) for o in losses]
        bg = mod.__dict__.get("_batch_graph")
        res[mode] = dict(losses=losses, issue=issue, ms_tail=1e3 * (t_end - t_half) / (a.steps - 1 - a.steps // 2),
                         replays=bg.replays if bg is not None else 0, failed=bg.failed if bg is not None else None,
                         flat_g=mod.optim_g.flat.clone(), flat_d=mod.optim_d.flat.clone())
        mod.optim_g.close()
        mod.optim_d.close()
        del mod
        ops.invalidate_weights()
        torch.cuda.empty_cache()
    graphed.set_batch_enabled(True)
    print("config %s / %s / %s / B=%d%s" % (a.config, a.workload, a.dtype, a.batch, " / forced 1-rank RCCL" if a.ddp else ""))
    print("graph loop: %d replays, failed=%s" % (res["graph"]["replays"], res["graph"]["failed"]))
    worst = 0.0
    for i, ((g0, d0), (g1, d1)) in enumerate(zip(res["eager"]["losses"], res["graph"]["losses"])):
        rg, rd = abs(g0 - g1) / abs(g0), abs(d0 - d1) / abs(d0)
        worst = max(worst, rg, rd)
        print("step %2d  eager g %.6f d %.6f | graph-loop g %.6f d %.6f | rel %.1e %.1e | issue ms eager %.2f graph-loop %.2f"
              % (i, g0, d0, g1, d1, rg, rd, 1e3 * res["eager"]["issue"][i], 1e3 * res["graph"]["issue"][i]))
    for k in ("flat_g", "flat_d"):
        d = (res["eager"][k] - res["graph"][k]).abs().max().item()
        print("%s: max |eager - graph-loop| after %d steps = %.3e (max |p| %.3e)" % (k, a.steps, d, res["eager"][k].abs().max().item()))
    print("device ms per batch over the second half: eager %.2f, graph-loop %.2f" % (res["eager"]["ms_tail"], res["graph"]["ms_tail"]))
    print("worst relative loss difference %.2e" % worst)
    if a.ddp:
        import torch.distributed as dist
        from vcvits_amd.light.optim import shutdown_flag_groups
        shutdown_flag_groups()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
