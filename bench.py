#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: training utterances/sec (one generator step + one
discriminator step per batch) at configs/base.json segment_size.

At N=1 the workload is BASELINE.json configs[1]: base widths, batch 16, fp32, HiFi-GAN
generator + MPD(8 periods + S) + MSD + STFT/mel-L1 step (`VocoderGAN`), synthetic
z_slice/waveform batches.  One "step" = one batch through both optimizer passes, AdamW included.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task description) with `roofline` (conv_gemm, the
dominant kernel, timed with HIP events on its own stream inside the timed region) and
`cpu_baseline` (the CPU oracle trainer on a bounded sample of the same workload)."""
import argparse
import copy
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
GFLOP_PER_UTT = {"vocoder": 400.0, "full": 488.0}  # SURVEY.md section 8d (4G + 9 D1, reference semantics)


def profiled_traffic():
    """HBM bytes per conv_dma_kernel launch from the latest committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json,
    produced by tools/profile_summary.py; PMC counters cannot be read from inside the benchmark process)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return round(d["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["vocoder", "full"], default="vocoder")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-launch event timing")
    return ap.parse_args()


CPU_THREADS = 16  # measured on the GPU box host (2 x EPYC 9575F, 256 hw threads): one B=1 oracle batch takes
#                   1.2 s at 16 torch threads, 1.9 s at 32, 5.1 s at 64 and ~545 s at 256 (oversubscription)


def cpu_baseline(cfg, workload, periods):
    """The oracle (CPU restatement, torch CPU autograd + torch.optim.AdamW) on a bounded sample of
    the same workload: batches of 2 utterances, 1 warm-up + 3 timed, at the thread count that is
    fastest on this host."""
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VCVITS, VocoderGAN
    cores = min(CPU_THREADS, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    module = (VocoderGAN if workload == "vocoder" else VCVITS)(**cfg)
    trainer = CpuTrainer(module.state_dict(), cfg, periods, vocoder_only=(workload == "vocoder"))
    del module
    m = cfg["model"]

    def make(seed):
        if workload == "vocoder":
            return synthetic.vocoder_batch(2, m["inter_channels"], seed=seed)
        batch = synthetic.full_batch(2, m["hubert_channels"], seed=seed)
        g = torch.Generator().manual_seed(seed)
        batch["noise"] = torch.randn(2, m["inter_channels"], 384, generator=g)
        batch["ids_slice"] = torch.tensor([10, 20])
        return batch

    trainer.batch(make(98))
    n = 3
    t0 = time.perf_counter()
    for i in range(n):
        trainer.batch(make(99 + i))
    dt = (time.perf_counter() - t0) / n
    return {"value": round(2.0 / dt, 4), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": "3 timed batches of 2 utterances after 1 warm-up (%s workload, fp32, torch-CPU oracle "
                      "with AdamW), %.2f s per batch on %d of %d host threads" % (workload, dt, cores,
                                                                                 os.cpu_count() or 1)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("VCVITS_FORCE_DDP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from vcvits_amd import _lib, configs, synthetic
    from vcvits_amd.light.vcvits import DEFAULT_PERIODS, VCVITS, VocoderGAN
    L = _lib.lib()

    cfg = configs.base()
    B = a.batch
    torch.manual_seed(1234)  # identical initial weights on every rank
    module = (VocoderGAN if a.workload == "vocoder" else VCVITS)(**cfg).to(dev)
    module.train()
    module.configure_optimizers()
    module.optim_g.broadcast_parameters()
    module.optim_d.broadcast_parameters()
    m = cfg["model"]
    if a.workload == "vocoder":
        batches = [synthetic.vocoder_batch(B, m["inter_channels"], seed=1234 + 17 * rank + i, device=dev)
                   for i in range(2)]
    else:
        batches = [synthetic.full_batch(B, m["hubert_channels"], seed=1234 + 17 * rank + i, device=dev)
                   for i in range(2)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        module.fit_batch(batches[i % 2])
    sync()
    prof = (not a.no_prof)
    if prof:
        _lib.check(L.vcv_prof_begin(4096 * max(a.steps, 1)), "vcv_prof_begin")
    t0 = time.perf_counter()
    for i in range(a.steps):
        module.fit_batch(batches[i % 2])
    sync()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    roof = None
    if prof:
        out = (ctypes.c_double * 12)()
        _lib.check(L.vcv_prof_end(out, 4), "vcv_prof_end")
        if os.environ.get("VCVITS_PROF_DUMP"):
            L.vcv_prof_dump(os.environ["VCVITS_PROF_DUMP"].encode())

        def cls(i, name):
            n, ms, fl = out[3 * i], out[3 * i + 1], out[3 * i + 2]
            if n <= 0 or ms <= 0:
                return None
            ach = fl / (ms * 1e-3) / 1e12
            return {"kernel": name, "achieved": round(ach, 2), "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                    "launches_per_step": n / a.steps, "avg_launch_us": round(1e3 * ms / n, 2),
                    "gflop_per_launch": round(fl / n / 1e9, 3), "share_of_step_time": round(ms * 1e-3 / dt, 3)}

        dma = cls(2, "conv_dma_kernel (fwd + dgrad + convT, LDS-DMA staging, all tile variants)")
        if dma:
            # dominant kernel by time; the other MFMA kernel families ride along for the record
            traffic, traffic_src = profiled_traffic()
            roof = {"bound": "mfma", "kernel": dma["kernel"], "achieved": dma["achieved"], "peak": FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": dma["frac"], "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                    "traffic_source": traffic_src,
                    "launches_per_step": dma["launches_per_step"], "avg_launch_us": dma["avg_launch_us"],
                    "gflop_per_launch": dma["gflop_per_launch"], "share_of_step_time": dma["share_of_step_time"],
                    "other_kernels": [k for k in (cls(3, "wgrad_dma_kernel"), cls(0, "conv_gemm_kernel (register-staged)"),
                                                   cls(1, "conv_wgrad_kernel (register-staged)")) if k]}
    if rank == 0:
        value = world * B * a.steps / dt
        line = {
            "metric": "training utterances/sec (gen+disc step) at base.json segment_size",
            "value": round(value, 3), "unit": "utterances/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs/base.json widths, HiFi-GAN generator + MPD(8 periods+S) + MSD + "
                                    "STFT/mel-L1, G step + D step + AdamW" if a.workload == "vocoder" else
                                    "configs/base.json full SynthesizerSVC (feature input) + MPD + MSD, G step + D step"),
                       "per_gpu_batch": B, "global_batch": B * world, "segment_size": 16384,
                       "parallelism": "dp%d" % world,
                       "algorithmic_gflop_per_utterance": GFLOP_PER_UTT[a.workload],
                       "algorithmic_tflops": round(value * GFLOP_PER_UTT[a.workload] / 1e3, 2)},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            periods = m.get("multi_period_discriminator_periods") or DEFAULT_PERIODS
            line["cpu_baseline"] = cpu_baseline(cfg, a.workload, periods)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
