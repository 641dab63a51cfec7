#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: training utterances/sec (one generator step + one
discriminator step per batch) at configs/base.json segment_size.

Default (N=1): BASELINE.json configs[1] -- base widths, batch 16, fp32, HiFi-GAN generator + MPD(8 periods + S) +
MSD + STFT/mel-L1 step (`VocoderGAN`), synthetic z_slice / waveform batches.  One "step" = one batch through both
optimizer passes, AdamW included.  The other BASELINE configurations are reachable with flags:

  configs[2]  --workload full --batch 32 --dtype bf16          (full SynthesizerSVC + MPD + MSD)
  configs[3]  --config 48k --workload full --dtype bf16        (per-GPU batch 16; the 8-GPU launch is the driver's)
  configs[4]  --config 48k --workload infer --dtype bf16       (flow inverse + decode, 64 x 10 s -> real-time factor)

  python bench.py --gpus 1 --steps 10 --warmup 3
  python bench.py --gpus N --steps K --warmup W     (N > 1 without torchrun: this process starts N rank processes
                                                      itself BEFORE touching the GPU, relays rank 0's line and exits
                                                      non-zero if fewer than N GPUs are visible or a rank fails --
                                                      the reference's own launch is one command too: train.sh:1,
                                                      train.py:98-100)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

The default invocation (N = 1, configs[1]) also times short legs of BASELINE configs[2] (full model, B = 32, bf16) and
configs[4] (48 kHz inference, 64 x 10 s, bf16) and of configs[3]'s per-rank workload (48 kHz full model, bf16, batch 16,
one rank) after the main timed region and reports them under `config.extra_configs` (`--no-extra` skips them).

Rank 0 prints ONE JSON line (contract in the task description) with `roofline` (the dominant kernel family, timed
with HIP events attached to each dispatch on the launch stream inside the timed region) and `cpu_baseline` (the CPU
oracle on a bounded sample of the same workload)."""
import argparse
import ctypes
import glob
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# MI355X_MICROARCH.md: dense MFMA peaks (no sparsity)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
# algorithmic GFLOP per utterance (4 G + 9 D1, reference semantics); infer: per 10 s utterance
GFLOP_PER_UTT = {("base", "vocoder"): 400.0, ("base", "full"): 488.0, ("48k", "vocoder"): 536.0, ("48k", "full"): 560.0,
                 ("48k", "infer"): 770.0, ("base", "infer"): 788.0}
# profiler classes of the library (csrc/prof.h) and the kernel families (rocprofv3 names) each one times
PROF_CLASSES = ["conv_gemm_kernel (register-staged)", "conv_wgrad_kernel (register-staged)",
                "packed-weight conv kernels: fwd + dgrad + convT (conv_x3_kernel, conv_pk_kernel<F32El | Bf16El>, conv_dma_kernel, resblock_pair_kernel)",
                "weight-gradient kernels (wgrad_dma_kernel, wgrad_bf16_kernel)",
                "fused attention kernels (rel_attn_fwd / bwd_rows / bwd_cols)"]
PROF_FAMILIES = [["conv_gemm_kernel"], ["conv_wgrad_kernel"], ["conv_x3_kernel", "conv_pk_kernel", "conv_dma_kernel", "resblock_pair_kernel"],
                 ["wgrad_dma_kernel", "wgrad_bf16_kernel"], ["rel_attn_fwd_kernel", "rel_attn_bwd_rows_kernel", "rel_attn_bwd_cols_kernel"]]
NCLS = len(PROF_CLASSES)


def kernel_source_hash():
    """sha256 over the kernel sources and their host dispatch: profiles recorded for another kernel set are stale."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "vcvits_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "vcvits_amd", "csrc", "*.h")) +
                    [os.path.join(ROOT, "include", "vcvits_hip.h")] + glob.glob(os.path.join(ROOT, "vcvits_amd", "ops", "*.py"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profiled_traffic(families, workload_key):
    """HBM bytes per launch of the kernel `families` of one profiler class (launch-weighted mean) from the newest
    committed rocprofv3 PMC passes (profiles/*_hbm_traffic.json, written by tools/profile_summary.py -- PMC counters
    cannot be read from inside this process).  Returns (bytes, source, stale): a file recorded for other kernel sources
    or another workload is reported as stale and its number withheld."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), key=os.path.getmtime)
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload", "base/vocoder/f32") != workload_key:
            continue
        fams = [d.get("kernels", {}).get(k) for k in families]
        fams = [k for k in fams if k]
        if not fams:
            continue
        src = os.path.relpath(f, ROOT)
        if d.get("kernel_source_hash") != kernel_source_hash():
            return None, src, True
        n = sum(k["launches_profiled"] for k in fams)
        return round(sum(k["launches_profiled"] * k["hbm_bytes_per_launch"] for k in fams) / n), src, False
    return None, None, False


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["base", "48k"], default="base")
    ap.add_argument("--workload", choices=["vocoder", "full", "infer"], default="vocoder")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 16; 64 for --workload infer)")
    ap.add_argument("--frames", type=int, default=938, help="--workload infer: frames per utterance (938 = 10 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-baseline-full", action="store_true",
                    help="skip the second CPU leg (SURVEY 8d's config 1: full model, B=2)")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-launch event timing")
    ap.add_argument("--no-host-probe", action="store_true",
                    help="skip the idle-device host-issue probe after the timed region (counter-collection passes: fewer steps)")
    ap.add_argument("--cpu-child", action="store_true", help=argparse.SUPPRESS)  # (internal: the CPU-baseline child process)
    ap.add_argument("--varlen", type=int, default=0, metavar="FRAMES",
                    help="--workload full: a stream of variable-length batches, padded lengths bucketed to multiples of FRAMES")
    ap.add_argument("--no-extra", action="store_true", help="skip the short configs[2] / configs[4] legs of the default run")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend (gloo only with --dry-run: launcher test on a CPU-only host)")
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, form the process group, run the timing collectives and print the line "
                         "without launching a kernel (tests the N-rank launcher where there is no GPU)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no torchrun around it
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a, argv):
    """Parent of an N-rank run.  Touches no GPU (torch.cuda.device_count() does not initialise HIP on this image):
    checks that N devices are visible, starts one fresh `bench.py` process per rank with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and returns non-zero if any rank failed.  Never falls back
    to fewer ranks."""
    import subprocess
    n = a.gpus
    if not a.dry_run and os.environ.get("VCVITS_BENCH_SKIP_DEVICE_CHECK") != "1":  # (the skip is for the launcher test)
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible; refusing to run fewer ranks\n" % (n, have))
            return 3
    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                "LOCAL_WORLD_SIZE": str(n), "VCVITS_BENCH_CHILD": "1"})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the optimizers' gloo side channel needs no hostname lookup
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    import tempfile
    procs = []
    out0_file = tempfile.TemporaryFile(mode="w+")  # (a file, not a pipe: the supervisor below polls instead of reading)
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=out0_file if r == 0 else subprocess.DEVNULL))
    # supervise: a rank that dies early (OOM, device fault) would leave its siblings in RCCL init / an all-reduce until
    # the watchdog fires minutes later -- when any child exits non-zero the others are terminated; overall limit
    # VCVITS_BENCH_TIMEOUT seconds (default 3600)
    deadline = time.time() + float(os.environ.get("VCVITS_BENCH_TIMEOUT", "3600"))
    failed = False
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes) or time.time() > deadline:
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    out0_file.seek(0)
    out0 = out0_file.read()
    out0_file.close()
    if failed and not any(codes):
        codes = [1] * n  # (timeout: every child was still healthy when it was stopped)
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if any(codes) or not lines:
        sys.stderr.write("bench.py: rank exit codes %s, %d JSON line(s) from rank 0\n" % (codes, len(lines)))
        sys.stdout.write(out0 or "")
        return 1
    print(lines[-1], flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------------
# Host threads for the CPU oracle.  torch's intra-op pool oversubscribes badly on the GPU box host (2 x EPYC 9575F,
# 256 hardware threads: a B=1 batch took 1.2 s at 16 threads and minutes at 256 when measured by hand in round 2), so
# the thread count is chosen by a short sweep measured IN this run and the sweep is reported as measured.
CPU_THREAD_CANDIDATES = (8, 16, 32, 64, 128)


def host_info():
    """What the CPU baseline ran on: hardware threads, the CPUs this process may use, the cgroup CPU quota, NUMA nodes and
    the OpenMP placement in effect -- so that a thread count like '16 of 256' can be read against the machine."""
    info = {"hardware_threads": os.cpu_count() or 1}
    try:
        info["sched_affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup_cpu_quota"] = open(path).read().strip()
            break
        except OSError:
            continue
    info["numa_nodes"] = len(glob.glob("/sys/devices/system/node/node[0-9]*")) or None
    info["omp_env"] = {k: os.environ[k] for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_NUM_THREADS") if k in os.environ}
    return info


def _cpu_trainer(cfg, workload, periods):
    from oracle.cpu_step import CpuTrainer
    from vcvits_amd import synthetic
    from vcvits_amd.light.vcvits import VCVITS, VocoderGAN
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
    return trainer, make


def _time_batches(trainer, make, n, seed0):
    t0 = time.perf_counter()
    for i in range(n):
        trainer.batch(make(seed0 + i))
    return (time.perf_counter() - t0) / n


def cpu_baseline(cfg, workload, periods, frames, also_full=False):
    """The oracle (CPU restatement: torch CPU autograd + torch.optim.AdamW) on a bounded sample of the same workload,
    at the torch thread count a sweep measured in this very run finds fastest."""
    hw = os.cpu_count() or 1
    if workload == "infer":
        cores = min(16, hw)
        torch.set_num_threads(cores)
        from oracle import vits_oracle as O
        from vcvits_amd.model.synthesizers.synthesizer_svc import SynthesizerSVC
        d, m = cfg["data"], cfg["model"]
        torch.manual_seed(0)
        net = SynthesizerSVC(d["filter_length"] // 2 + 1, 32, n_speakers=d["n_speakers"], **m)
        sd = {"n." + k: v.detach() for k, v in net.state_dict().items()}
        C, H, T = m["inter_channels"], m["hidden_channels"], max(8, frames // 10)
        z_p = torch.randn(1, C, T)
        g = torch.randn(1, m["gin_channels"], 1)
        mask = torch.ones(1, 1, T)

        def once():
            with torch.no_grad():
                z = O.flow_forward(sd, "n.flow", z_p, mask, g, True, C, H, 5, 1, 4)
                return O.generator_forward(sd, "n.dec", z * mask, m["upsample_rates"], m["upsample_kernel_sizes"],
                                           m["resblock_kernel_sizes"], m["resblock_dilation_sizes"])
        once()
        t0 = time.perf_counter()
        o = once()
        dt = time.perf_counter() - t0
        audio = o.shape[-1] / d["target_sampling_rate"]
        return {"value": round(dt / audio, 5), "unit": "wall s / audio s", "cores": cores, "kind": "port",
                "sample": "1 utterance x %d frames (%.2f s of audio) after 1 warm-up: oracle flow reverse + HiFi-GAN "
                          "decode, fp32 torch-CPU, %.2f s" % (T, audio, dt),
                "host": dict(host_info(), torch_threads_used=cores)}
    trainer, make = _cpu_trainer(cfg, workload, periods)
    sweep = {}
    for th in sorted(set(min(c, hw) for c in CPU_THREAD_CANDIDATES)):
        torch.set_num_threads(th)
        if not sweep:
            trainer.batch(make(98))  # warm-up (allocator, oneDNN primitive caches)
        sweep[th] = _time_batches(trainer, make, 1, 90 + th)
        if sweep[th] > 1.6 * min(sweep.values()):
            break  # (bounded: past the knee every doubling of the thread count has been slower still)
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    dt = _time_batches(trainer, make, 3, 99)
    out = {"value": round(2.0 / dt, 4), "unit": "utterances/s", "cores": cores, "kind": "port",
           "sample": "3 timed batches of 2 utterances after 1 warm-up (%s workload, fp32, torch-CPU oracle with AdamW), "
                     "%.2f s per batch on %d of %d host threads" % (workload, dt, cores, hw),
           "host": dict(host_info(), torch_threads_used=cores,
                        thread_sweep_measured_s_per_B2_batch={str(k): round(v, 3) for k, v in sweep.items()})}
    if also_full and workload != "full":
        # SURVEY 8d's CPU baseline proper: BASELINE configs[0] (full model, B=2, one G step + one D step)
        trainer, make = _cpu_trainer(cfg, "full", periods)
        trainer.batch(make(98))
        dtf = _time_batches(trainer, make, 2, 99)
        out["config1_full_model_B2"] = {"value": round(2.0 / dtf, 4), "unit": "utterances/s", "s_per_batch": round(dtf, 2)}
    return out


def cpu_baseline_subprocess(config, workload, frames, also_full):
    """The CPU baseline in a fresh CHILD process (started, not exec'ed: this process holds the GPU) that never touches the
    GPU: its thread pools and allocator are its own, and `host` reports what the container really grants (CPUs the
    process may run on, cgroup CPU quota, NUMA nodes).  Falls back to measuring in-process if the child fails."""
    import subprocess
    env = dict(os.environ)
    # (OMP_PROC_BIND=close / OMP_PLACES=cores were tried here: 8 / 16 bound threads took 1.14 / 1.29 s per batch on a box where
    # 16 unbound threads take 0.93 s -- the container's cgroup grants 16 CPUs of quota (`host.cgroup_cpu_quota`) wherever the
    # scheduler likes to place them, which is also why more than 16 threads only add throttling)
    env.pop("OMP_NUM_THREADS", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-child", "--config", config, "--workload", workload,
           "--frames", str(frames)] + ([] if also_full else ["--no-cpu-baseline-full"])
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            out = json.loads(lines[-1])
            out["measured_in"] = "child process (no GPU context)"
            return out
        sys.stderr.write("bench.py: CPU-baseline child failed (rc %s): %s\n" % (r.returncode, r.stderr[-400:]))
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("bench.py: CPU-baseline child failed: %s\n" % e)
    return None


# ---------------------------------------------------------------------------------------------------------------
# one timed leg
# ---------------------------------------------------------------------------------------------------------------
def build_infer(cfg, B, T, dev):
    from vcvits_amd.model.synthesizers.synthesizer_svc import SynthesizerSVC
    d, m = cfg["data"], cfg["model"]
    net = SynthesizerSVC(d["filter_length"] // 2 + 1, 32, n_speakers=d["n_speakers"], **m).to(dev).eval()
    with torch.no_grad():
        for p in net.flow.parameters():
            if p.abs().sum() == 0:
                p.normal_(0, 0.02)  # the zero-initialised `post` convs would make the flow an identity
    g = torch.Generator().manual_seed(1234)
    m_p = torch.randn(B, m["inter_channels"], T, generator=g).to(dev)
    logs_p = (torch.randn(B, m["inter_channels"], T, generator=g) * 0.1 - 1.0).to(dev)
    noise = torch.randn(B, m["inter_channels"], T, generator=g).to(dev)
    y_mask = torch.ones(B, 1, T, device=dev)
    spk = net.emb_g(torch.randint(0, d["n_speakers"], (B,), generator=g).to(dev)).unsqueeze(-1)
    from vcvits_amd import ops

    def step():
        with torch.no_grad():
            z_p = ops.prior_sample(m_p, logs_p, noise, 1.0)
            z = net.flow(z_p, y_mask, g=spk, reverse=True)
            return net.dec(ops.mask_mul(z, y_mask.reshape(B, -1)))
    return step


def varlen_batches(cfg, B, n, seed, dev):
    """`n` collate-shaped batches of the full-model workload whose utterance lengths vary (spectrogram frames uniform in
    [256, 384], content frames 0.53 of them -- SURVEY 8d's 384 / 204 at the top), each padded to ITS OWN longest utterance as
    the reference collate does (vits/data/collate.py:133-190): raw shapes practically never repeat."""
    from vcvits_amd import synthetic
    m, hop = cfg["model"], cfg["data"]["hop_length"]
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        ty = torch.randint(256, 385, (B,), generator=g)
        tx = (ty.float() * 0.53).long().clamp(min=8)
        b = synthetic.full_batch(B, m["hubert_channels"], t_y=int(ty.max()), t_x=int(tx.max()), seed=seed + 1 + i)
        for r in range(B):
            b["y_wav_lengths"][r] = int(ty[r]) * hop
            b["y_wav_values"][r, :, int(ty[r]) * hop:] = 0.0
            b["x_hubert_features_lengths"][r] = b["x_pitch_lengths"][r] = int(tx[r])
            b["x_hubert_features_values"][r, :, int(tx[r]):] = 0.0
            b["x_pitch_values"][r, int(tx[r]):] = 0
        out.append({k: v.to(dev) for k, v in b.items()})
    return out


def run_leg(config, workload, dtype, batch, frames, steps, warmup, dev, world, rank, prof=True, f32_split=None, host_probe=True,
            varlen=0):
    """Build the workload, do `warmup` untimed steps, time exactly `steps` steps between barrier + synchronize on both
    sides, MAX over ranks.  Returns the pieces of the JSON line."""
    from vcvits_amd import _lib, configs, ops, synthetic
    from vcvits_amd.light.vcvits import DEFAULT_PERIODS, VCVITS, VocoderGAN
    L = _lib.lib()
    ops.set_compute_dtype(dtype)
    if f32_split is not None:  # (on, terms): the fp32 arithmetic variant of this leg; default = the library's default
        ops.set_f32_split(f32_split[0], terms=f32_split[1])
    infer = workload == "infer"
    cfg = configs.base() if config == "base" else configs.base_48k()
    B = batch if batch is not None else (64 if infer else 16)
    m = cfg["model"]
    torch.manual_seed(1234)  # identical initial weights on every rank
    module = None
    if infer:
        run = build_infer(cfg, B, frames, dev)
    else:
        if varlen:
            cfg["train"]["length_bucket_frames"] = int(varlen)  # (data/collate.py: padded lengths rounded up to multiples)
        module = (VocoderGAN if workload == "vocoder" else VCVITS)(**cfg).to(dev)
        module.train()
        module.configure_optimizers()
        module.optim_g.broadcast_parameters()
        module.optim_d.broadcast_parameters()
        make = synthetic.vocoder_batch if workload == "vocoder" else synthetic.full_batch
        width = m["inter_channels"] if workload == "vocoder" else m["hubert_channels"]
        if varlen:
            batches = varlen_batches(cfg, B, 16, 4321 + 17 * rank, dev)
        else:
            batches = [make(B, width, seed=1234 + 17 * rank + i, device=dev) for i in range(2)]

        def run(i=[0]):
            module.fit_batch(batches[i[0] % len(batches)])
            i[0] += 1

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, before the W warm-up steps: the batch graph is RECORDED here (light/graphed.py records a batch shape the third
    # time it sees it) -- part of building the workload, like a compile step, so that a small W does not put the one-off
    # capture inside the timed region; with the default W = 3 it changes nothing (the third warm-up step was the capture)
    setup_steps = 0
    if module is not None and varlen:
        # every bucketed shape of the stream is seen (and recorded on its third sighting) before the warm-up: three rounds
        for _ in range(3 * len(batches)):
            run()
            setup_steps += 1
        sync()
    elif module is not None:
        # (data parallel: recording waits until both optimizers have frozen their used-parameter sets -- two steps of
        # cross-rank agreement -- and then for the third sighting of the shape: up to eight setup steps instead of four,
        # every rank taking the same number)
        ddp_opts = [o for o in (module.optim_g, module.optim_d) if getattr(o, "_ddp", False)]
        max_setup = 8 if ddp_opts else 4
        while setup_steps < max_setup:
            bg = module.__dict__.get("_batch_graph")
            if bg is not None:
                if bg.replays > 0:
                    break
                if not bg.applicable():
                    waiting = bool(ddp_opts) and not bg.failed and any(o._static_set is None for o in ddp_opts)
                    if not waiting:
                        break
            run()
            setup_steps += 1
        sync()
    if module is not None and prof:
        # still setup: the timed region mixes replayed batches with eager ones (every fourth carries per-launch events), and
        # recording a batch EMPTIES the caching allocator (torch's capture entry) -- the first eager batch after it would pay
        # for ~10 GB of fresh hipMallocs inside the timed region (seen once as a 130-300 ms first step on a fresh box).  One
        # eager batch here gives the eager loop its pools back.
        from vcvits_amd.light import graphed as _graphed
        bg = module.__dict__.get("_batch_graph")
        if bg is not None and bg.replays > 0:
            was = _graphed.BATCH_ENABLED[0]
            _graphed.set_batch_enabled(False)
            run()
            _graphed.set_batch_enabled(was)
            setup_steps += 1
            sync()
    for _ in range(warmup):
        run()
    sync()
    if prof:
        _lib.check(L.vcv_prof_begin(8192 * max(steps, 1)), "vcv_prof_begin")
    # per-launch events on every fourth step of the timed region (steps 0, 4, ...): dispatch-attached events cost ~4 % of
    # a step (consecutive kernels no longer overlap their launch latencies), so the measurement samples instead of taking
    # that from the measured value; the roofline figures are per profiled step (383 launches of the dominant class each)
    PEVERY = 4
    psteps = (steps + PEVERY - 1) // PEVERY if prof else 0
    calls0 = _lib.CALLS[0]
    bg = module.__dict__.get("_batch_graph") if module is not None else None
    replays0 = bg.replays if bg is not None else 0
    t_replay = t_eager = 0.0
    n_replay = n_eager = 0
    c0 = time.thread_time()
    t0 = time.perf_counter()
    for i in range(steps):
        if prof:
            L.vcv_prof_pause(0 if i % PEVERY == 0 else 1)
        ts = time.perf_counter()
        run()
        te = time.perf_counter() - ts
        if prof and i % PEVERY == 0:  # (the profiled steps run eagerly: per-launch events are not capturable)
            t_eager, n_eager = t_eager + te, n_eager + 1
        else:
            t_replay, n_replay = t_replay + te, n_replay + 1
    t_issue = time.perf_counter() - t0  # host time to ISSUE the timed steps (no device wait inside a step)
    c_issue = time.thread_time() - c0   # ... and the CPU time this thread spent doing it
    sync()
    dt = time.perf_counter() - t0
    calls = (_lib.CALLS[0] - calls0) / max(steps, 1)
    bg = module.__dict__.get("_batch_graph") if module is not None else None
    n_graph = (bg.replays - replays0) if bg is not None else 0
    # Host time to ISSUE one step with the device idle -- what the host itself costs.  Inside the timed region the host runs
    # ahead of the GPU until the launch queue is full and then blocks at the device's pace (a replayed batch is ~1,000-1,800
    # queued packets), so the wall time per issue there measures back-pressure, not host work; both are reported.
    unloaded = {}
    if module is not None and host_probe:
        from vcvits_amd.light import graphed

        def issue_once():
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run()
            d = time.perf_counter() - t1
            torch.cuda.synchronize()
            return 1e3 * d
        if prof:
            L.vcv_prof_pause(1)
        if n_graph > 0:
            unloaded["graph"] = min(issue_once() for _ in range(3))
        was = graphed.BATCH_ENABLED[0]
        graphed.set_batch_enabled(False)
        unloaded["eager"] = min(issue_once() for _ in range(2))
        graphed.set_batch_enabled(was)
    n_eager_steps = steps - n_graph
    if unloaded:
        mix = (n_graph * unloaded.get("graph", 0.0) + n_eager_steps * unloaded["eager"]) / max(steps, 1)
    else:
        mix = 1e3 * t_issue / max(steps, 1)
    host = {"issue_ms": mix, "cpu_ms": 1e3 * c_issue / max(steps, 1), "graph_replays": n_graph,
            "wall_ms_in_timed_region": 1e3 * t_issue / max(steps, 1),
            "issue_ms_unloaded_graph_step": round(unloaded["graph"], 2) if "graph" in unloaded else None,
            "issue_ms_unloaded_eager_step": round(unloaded["eager"], 2) if "eager" in unloaded else None,
            "wall_ms_unprofiled_step": round(1e3 * t_replay / n_replay, 2) if n_replay else None,
            "wall_ms_profiled_eager_step": round(1e3 * t_eager / n_eager, 2) if n_eager else None}
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    peak = PEAK_TFLOPS[dtype]
    roof = None
    if prof:
        out = (ctypes.c_double * (3 * NCLS))()
        nbytes = (ctypes.c_double * NCLS)()
        _lib.check(L.vcv_prof_end(out, NCLS), "vcv_prof_end")
        _lib.check(L.vcv_prof_bytes(nbytes, NCLS), "vcv_prof_bytes")
        roofs = (ctypes.c_double * NCLS)()
        _lib.check(L.vcv_prof_roof(roofs, NCLS), "vcv_prof_roof")
        if os.environ.get("VCVITS_PROF_DUMP"):
            L.vcv_prof_dump(os.environ["VCVITS_PROF_DUMP"].encode())

        def cls(i):
            n, ms, fl = out[3 * i], out[3 * i + 1], out[3 * i + 2]
            if n <= 0 or ms <= 0:
                return None
            ach = fl / (ms * 1e-3) / 1e12
            # roofline of a class whose launches run on different matrix pipes (fp32-input MFMA 157.3 TFLOP/s, bf16 MFMA
            # 2.5 PFLOP/s; split-operand fp32 launches execute 6 or 9 bf16 products per fp32 product): time at the dense
            # peak of each launch's own pipe, summed, over the measured time; `peak` = the algorithmic rate that sum allows
            frac = roofs[i] / (ms * 1e-3)
            return {"kernel": PROF_CLASSES[i], "achieved": round(ach, 2), "frac": round(frac, 4),
                    "peak": round(fl / roofs[i] / 1e12, 1),
                    "launches_per_step": n / psteps, "avg_launch_us": round(1e3 * ms / n, 2),
                    "gflop_per_launch": round(fl / n / 1e9, 3), "share_of_step_time": round(ms * 1e-3 / (dt * psteps / steps), 3),
                    "algorithmic_bytes_per_launch": round(nbytes[i] / n) if nbytes[i] > 0 else None}

        fams = [c for c in (cls(i) for i in range(NCLS)) if c]
        if fams:
            # the dominant kernel family by time carries the roofline; the others ride along for the record
            dom = max(fams, key=lambda c: c["share_of_step_time"])
            traffic, traffic_src, stale = profiled_traffic(PROF_FAMILIES[PROF_CLASSES.index(dom["kernel"])],
                                                           "%s/%s/%s" % (config, workload, dtype))
            roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": dom["peak"],
                    "unit": "TFLOP/s", "frac": dom["frac"],
                    # for reference only: the same achieved rate against the dtype's own MFMA peak (fp32-input MFMA 157.3)
                    "frac_vs_dtype_mfma_peak": round(dom["achieved"] / peak, 4),
                    "peak_note": "algorithmic TFLOP/s at the dense MFMA peak of the pipe each launch of the class runs on "
                                 "(fp32-input MFMA 157.3; bf16 MFMA 2500, / 6 or / 9 for split-operand fp32 launches); "
                                 "random-data bf16 MFMA loops sustain ~1250 of the 2500 on this chip (clock give-back, "
                                 "MI355X_MICROARCH.md)", "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                    "traffic_source": traffic_src, "traffic_stale": stale,
                    "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                    "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_launch_us"],
                    "gflop_per_launch": dom["gflop_per_launch"], "share_of_step_time": dom["share_of_step_time"],
                    "profiled_steps": "%d of the %d timed steps (every fourth)" % (psteps, steps),
                    "kernel_source_hash": kernel_source_hash(),
                    "other_kernels": [c for c in fams if c is not dom]}
    periods = m.get("multi_period_discriminator_periods") or DEFAULT_PERIODS
    vstats = None
    if varlen and bg is not None:
        vstats = {"bucket_frames": int(varlen), "batches_in_stream": len(batches),
                  "distinct_raw_shapes": len({tuple(b["y_wav_values"].shape) + tuple(b["x_pitch_values"].shape) for b in batches}),
                  "recorded_graphs": len(bg.entries), "graph_pool_bytes": bg.held_bytes(), "captures": bg.captures,
                  "evictions": bg.evictions, "out_of_memory_captures": bg.ooms}
    if module is not None:
        module.optim_g.close()
        module.optim_d.close()
        module.drop_graphs()  # (the recorded batches own the activation footprint of their shapes: give it back before the next leg)
        batches = None
    bg = None
    del run, module
    ops.invalidate_weights()
    ops.set_compute_dtype("f32")
    arith = f32_arithmetic(ops, L) if dtype == "f32" else None
    if f32_split is not None:
        ops.set_f32_split(True, terms=6)
    torch.cuda.empty_cache()
    return {"varlen": vstats, "arith": arith, "dt": dt, "roof": roof, "host_issue_ms": host["issue_ms"], "host": host, "calls": calls, "B": B, "cfg": cfg, "periods": periods, "config": config, "workload": workload,
            "dtype": dtype, "frames": frames, "steps": steps, "warmup": warmup, "world": world, "setup_steps": setup_steps}


def f32_arithmetic(ops, L):
    """How the fp32 GEMM-shaped launches of the leg just run were computed."""
    if not ops._USE_X3[0]:
        return "fp32 throughout: fp32-input MFMA (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain)"
    n = L.vcv_conv_x3_get_terms()
    return ("fp32 throughout; GEMM-shaped launches by exact operand splitting: every fp32 operand = 3 bf16 terms, %d of the 9 "
            "bf16 MFMA products per fp32 product (%s), fp32 accumulate; the rest on fp32-input MFMA / fp32 VALU"
            % (n, "all: exact operands" if n == 9 else "the 3 left out are each < 2^-24 of the product; error vs float64 "
                                                        "equal to the 9-term and the fmaf-chain kernels': profiles/r3_x3_vs_f64.txt"))


def make_line(r):
    """The JSON line of one leg (contract in the task description)."""
    cfg, B, world, steps, dt, frames = r["cfg"], r["B"], r["world"], r["steps"], r["dt"], r["frames"]
    config, workload, dtype = r["config"], r["workload"], r["dtype"]
    infer = workload == "infer"
    gfl = GFLOP_PER_UTT[(config, workload)] * (frames / 938.0 if infer else 1.0)
    utt_s = world * B * steps / dt
    cfgname = "configs/base.json" if config == "base" else "configs/48k_base.json"
    if infer:
        audio_s = B * frames * cfg["data"]["hop_length"] / cfg["data"]["target_sampling_rate"]
        line = {"metric": "inference real-time factor (infer.py voice-conversion path: flow inverse + HiFi-GAN decode)",
                "value": round(dt / steps / audio_s / world, 7), "unit": "wall s / audio s",
                "higher_is_better": False, "scaling": "weak"}
        wl = "%s widths, prior sample + flow reverse + HiFi-GAN decode, %d x %d frames (%.1f s each)" % (
            cfgname, B, frames, frames * cfg["data"]["hop_length"] / cfg["data"]["target_sampling_rate"])
    else:
        line = {"metric": "training utterances/sec (gen+disc step) at base.json segment_size",
                "value": round(utt_s, 3), "unit": "utterances/s", "higher_is_better": True, "scaling": "weak"}
        wl = ("%s widths, HiFi-GAN generator + MPD(%d periods+S) + MSD + STFT/mel-L1, G step + D step + AdamW"
              % (cfgname, len(r["periods"])) if workload == "vocoder" else
              "%s full SynthesizerSVC (feature input) + MPD + MSD, G step + D step + AdamW" % cfgname)
    line.update({"n_gpus": world, "steps": steps, "warmup": r["warmup"], "ms_per_step": round(1e3 * dt / steps, 3),
                 "vs_baseline": None, "dtype": dtype, "data": "synthetic",
                 "config": {"workload": wl, "per_gpu_batch": B, "global_batch": B * world,
                            "segment_size": cfg["train"]["segment_size"],
                            "parallelism": ("dp%d" % world) if not infer else ("replicas%d" % world),
                            "process_group_ranks": dist.get_world_size() if dist.is_initialized() else 1,
                            "utterances_per_s": round(utt_s, 3), "algorithmic_gflop_per_utterance": round(gfl, 1),
                            "algorithmic_tflops": round(utt_s * gfl / 1e3, 2),
                            # the figure counts the reference's work (4 generator forwards-equivalents + 9 D); the
                            # discriminator step's no-grad generator pass here is decoder-side only (posterior encoder ->
                            # slice -> decoder: what y_hat depends on), so the EXECUTED flops of a full-model step are lower
                            **({"algorithmic_gflop_note": "reference semantics (SURVEY 8d: 4 G + 9 D1); the D-step's no-grad "
                                "generator pass skips the content encoder and the flow (dead work for y_hat.detach()), "
                                "executed GFLOP per utterance are lower by that pass's enc_p + flow share"}
                               if workload == "full" else {}),
                            # host side of a step: time this process needed to ISSUE one step's launches (the step is
                            # GPU-bound while this stays below ms_per_step) and the library launcher calls it made
                            # host side of a step.  A training batch whose shapes repeat is ONE HIP-graph replay
                            # (vcvits_amd/light/graphed.py); the steps that carry per-launch events (every fourth) run the
                            # eager loop.  host_issue_ms_per_step = host time to issue a step with the device idle (measured
                            # after the timed region: graph replay / eager step), weighted by the timed region's mix of the
                            # two; host_wall_ms_per_step_in_timed_region is the wall time the issuing thread spent per step
                            # inside the timed region, where it runs ahead until the launch queue is full and then waits for
                            # the device (back-pressure, not host work)
                            "host_issue_ms_per_step": round(r["host_issue_ms"], 2),
                            "host_issue_ms_graph_replay_step": r["host"]["issue_ms_unloaded_graph_step"],
                            "host_issue_ms_eager_step": r["host"]["issue_ms_unloaded_eager_step"],
                            "hip_graph_replays_in_timed_steps": r["host"]["graph_replays"],
                            "graph_recording_steps_before_warmup": r["setup_steps"],
                            "host_wall_ms_per_step_in_timed_region": round(r["host"]["wall_ms_in_timed_region"], 2),
                            "host_cpu_ms_per_step_in_timed_region": round(r["host"]["cpu_ms"], 2),
                            "library_launcher_calls_per_step": round(r["calls"], 1),
                            "arithmetic": (r["arith"] if dtype == "f32" else
                                           "bf16 MFMA operands, fp32 accumulate, fp32 master weights / losses / optimizer")},
                 "roofline": r["roof"]})
    if r.get("varlen"):
        line["config"]["variable_length_stream"] = r["varlen"]
        line["config"]["workload"] += ("; 16 batches of VARIABLE utterance lengths (256-384 frames), padded lengths bucketed to "
                                       "multiples of %d frames (train.length_bucket_frames)" % r["varlen"]["bucket_frames"])
    return line


def short(line):
    """An extra leg as it is carried inside the main line."""
    roof = line.get("roofline") or {}
    return {"metric": line["metric"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
            "steps": line["steps"], "warmup": line["warmup"], "dtype": line["dtype"],
            "workload": line["config"]["workload"], "per_gpu_batch": line["config"]["per_gpu_batch"],
            "algorithmic_tflops": line["config"]["algorithmic_tflops"],
            "host_issue_ms_per_step": line["config"]["host_issue_ms_per_step"],
            "host_issue_ms_graph_replay_step": line["config"].get("host_issue_ms_graph_replay_step"),
            "hip_graph_replays_in_timed_steps": line["config"].get("hip_graph_replays_in_timed_steps"),
            "library_launcher_calls_per_step": line["config"]["library_launcher_calls_per_step"],
            **({"variable_length_stream": line["config"]["variable_length_stream"]} if "variable_length_stream" in line["config"] else {}),
            "roofline": {k: roof.get(k) for k in ("kernel", "achieved", "peak", "frac", "traffic", "traffic_source",
                                                  "avg_launch_us", "launches_per_step", "share_of_step_time",
                                                  "algorithmic_bytes_per_launch")} if roof else None}


def dry_run(a, world, rank):
    """Launcher test: N ranks, process group, the timing collectives, one line -- no kernels, no GPU."""
    dist.init_process_group(a.backend, rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dry run (launcher only)", "value": 0.0, "unit": "utterances/s", "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * float(tt) / max(a.steps, 1), 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
                          "data": "none", "config": {"workload": "dry run", "process_group_ranks": dist.get_world_size(),
                                                     "backend": a.backend}}), flush=True)
    dist.destroy_process_group()


def pin_rank_cpus(local_rank, local_world):
    """Give each local rank its own contiguous slice of the CPUs this process may run on (one NUMA-local block per GPU on
    the usual 2-socket / 8-GPU node, where GPUs 0-3 hang off socket 0) and size the OpenMP / torch intra-op pools to it.
    Called BEFORE anything initialises the GPU.  VCVITS_BENCH_NO_PIN=1 leaves the scheduler alone."""
    if local_world <= 1 or os.environ.get("VCVITS_BENCH_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // local_world
        if per < 1:
            return None
        mine = cpus[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(per, 16)))
        return mine
    except OSError:
        return None


def preflight(dev, world, rank):
    """Before anything is built or timed on an N-rank run: every rank sees N devices and sits on its own, the RCCL group
    moves one 1 MB all-reduce with the right sum, and all ranks agree on it.  A node that cannot do this fails HERE with a
    message naming the rank and the reason, not minutes later inside the first gradient bucket."""
    have = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if have < local_world:
        raise SystemExit("bench.py preflight: rank %d sees %d GPU(s), the launch needs %d on this node" % (rank, have, local_world))
    t = torch.full((262144,), float(rank + 1), device=dev, dtype=torch.float32)  # 1 MB
    t0 = time.perf_counter()
    dist.all_reduce(t)
    torch.cuda.synchronize()
    want = world * (world + 1) / 2.0
    got = float(t[0]), float(t[-1])
    if got != (want, want):
        raise SystemExit("bench.py preflight: rank %d: 1 MB all-reduce returned %r, expected %r" % (rank, got, want))
    # every rank reports which device it holds; rank 0 checks that they are all different
    ids = torch.zeros(world, device=dev, dtype=torch.int64)
    ids[rank] = torch.cuda.current_device() + 1
    dist.all_reduce(ids)
    torch.cuda.synchronize()
    if world == local_world and len(set(ids.tolist())) != world:
        raise SystemExit("bench.py preflight: ranks share devices: %s" % ids.tolist())
    if rank == 0:
        sys.stderr.write("bench.py preflight: %d ranks, %d devices visible, 1 MB all-reduce ok (%.1f ms incl. RCCL init)\n"
                         % (world, have, 1e3 * (time.perf_counter() - t0)))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse(argv)
    if a.cpu_child:  # CPU only: the oracle on the host cores, one JSON object on stdout
        from vcvits_amd import configs
        from vcvits_amd.light.vcvits import DEFAULT_PERIODS
        cfg = configs.base() if a.config == "base" else configs.base_48k()
        periods = cfg["model"].get("multi_period_discriminator_periods") or DEFAULT_PERIODS
        print(json.dumps(cpu_baseline(cfg, a.workload, periods, a.frames, also_full=not a.no_cpu_baseline_full)), flush=True)
        return
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        # no torchrun around us: become the launcher (nothing has touched the GPU in this process)
        sys.exit(launch_ranks(a, argv))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and (a.gpus > 1 or world > 1):
        raise SystemExit("bench.py: --gpus %d disagrees with WORLD_SIZE=%d" % (a.gpus, world))
    pinned = pin_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if a.dry_run:
        return dry_run(a, world, rank)
    if a.backend != "nccl":
        raise SystemExit("bench.py: --backend gloo is for --dry-run only (the product path is RCCL)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    infer = a.workload == "infer"
    if world > 1 or os.environ.get("VCVITS_FORCE_DDP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        # training: gradient all-reduce over RCCL; inference: replicas, the group carries the timing barrier only
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != world:
            raise SystemExit("bench.py: process group has %d ranks, expected %d" % (dist.get_world_size(), world))
        preflight(dev, world, rank)
    r = run_leg(a.config, a.workload, a.dtype, a.batch, a.frames, a.steps, a.warmup, dev, world, rank, prof=not a.no_prof,
                host_probe=not a.no_host_probe, varlen=a.varlen if a.workload == "full" else 0)
    line = make_line(r) if rank == 0 else None
    default_run = (world == 1 and a.config == "base" and a.workload == "vocoder" and a.dtype == "f32"
                   and a.batch is None and not a.no_extra)
    if default_run:
        # BASELINE configs[2] and configs[4], a few steps each, so that the driver's line carries them too
        extra = {}
        legs = {"configs[2]": ("base", "full", "bf16", 32, None),
                # configs[3]'s per-rank workload (48 kHz full model, bf16, per-GPU batch 16) on this ONE rank: what each of
                # the eight ranks computes between its gradient all-reduces; the 8-GPU line itself is `--gpus 8`
                "configs[3], one rank (per-GPU batch 16; no collective)": ("48k", "full", "bf16", 16, None),
                "configs[4]": ("48k", "infer", "bf16", 64, None),
                # configs[2] on batches as a real filelist gives them: variable utterance lengths, each batch padded to its own
                # longest by the reference collate; bucketed to 64-frame multiples so that shapes repeat and batches replay
                "configs[2], variable-length batches (256-384 frames, bucketed to 64)": ("base", "full", "bf16", 32, None),
                # the headline workload in the two other fp32 arithmetics the library offers
                "configs[1], nine product terms (exact operands)": ("base", "vocoder", "f32", None, (True, 9)),
                "configs[1], fp32-input MFMA kernels (fmaf chain, no operand splitting)": ("base", "vocoder", "f32", None, (False, None))}
        for key, (c, w, dt_, b, split) in legs.items():
            try:
                if key.startswith("configs[2], variable"):
                    ln = make_line(run_leg(c, w, dt_, b, 938, 16, 3, dev, 1, 0, prof=not a.no_prof, varlen=64))
                    extra[key] = short(ln)
                    continue
                # (10 timed steps after 3 warm-ups, like the headline leg: with 4 + 2 the first-time packs / plans /
                # allocator growth of a fresh model were still inside the timed region on some boxes)
                ln = make_line(run_leg(c, w, dt_, b, 938, 10, 3, dev, 1, 0, prof=not a.no_prof, f32_split=split))
                extra[key] = short(ln)
                if split is not None:
                    extra[key]["arithmetic"] = ln["config"]["arithmetic"]
            except Exception as e:  # a failed extra leg must not cost the headline line
                extra[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        line["config"]["extra_configs"] = extra
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = (cpu_baseline_subprocess(a.config, a.workload, a.frames, not a.no_cpu_baseline_full)
                                    or cpu_baseline(r["cfg"], a.workload, r["periods"], a.frames,
                                                    also_full=not a.no_cpu_baseline_full))
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    # a stuck run can be asked where it is: SIGUSR1 writes every thread's Python stack to stderr (tools/run_profiles.sh)
    import faulthandler
    import signal
    faulthandler.register(signal.SIGUSR1, all_threads=True)
    main()
