"""Which launches of the benchmarked workloads still land on the library's three general-purpose kernels -- the register-staged
implicit-GEMM (conv_gemm.hip: `vcv_conv_gemm`), its weight-gradient twin (conv_wgrad.hip: `vcv_conv_wgrad`) and the LDS-DMA
kernel of round 1 (conv_dma.hip: `vcv_conv_dma_run`) -- with their shapes and call counts per training step / decode.
Output: profiles/r6_fallback_census.txt (DESIGN.md section 4 cites it).

  python3 tools/fallback_census.py            (on the GPU box)"""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vcvits_amd import _lib, configs, ops, synthetic  # noqa: E402
from vcvits_amd.light import graphed  # noqa: E402
from vcvits_amd.light.vcvits import VCVITS, VocoderGAN  # noqa: E402

TF = {0: "-", 1: "leaky", 2: "dleaky", 3: "drelu", 4: "dtanh", 5: "dlogclamp"}
ACT = {0: "-", 1: "leaky", 2: "relu", 3: "tanh", 4: "logclamp"}


def census(run, steps):
    L = _lib.lib()
    stat = collections.Counter()
    orig = {n: getattr(L, n) for n in ("vcv_conv_gemm", "vcv_conv_wgrad", "vcv_conv_dma_run")}

    def conv_key(name, a):
        kind = "fwd" if a.a_mode == 0 else ("dgrad/convT x%d phases" % a.phases if a.phases > 1 else "dgrad/convT")
        return (name, "%s G=%d %d->%d K=%d s=%d dil=%d rows=%d x P=%d B=%d in_tf=%s act=%s out_tf=%s%s%s" % (
            kind, a.G, a.Cg * a.G, a.Mg * a.G, a.K, a.s, abs(a.dj), a.Tout, a.P, a.B, TF.get(a.in_tf, a.in_tf), ACT.get(a.out_act, a.out_act),
            TF.get(a.out_tf, a.out_tf), " +res" if a.res else "", " +mask" if a.mask else ""))

    def gemm(ap, st):
        stat[conv_key("conv_gemm_kernel", ctypes.cast(ap, ctypes.POINTER(_lib.VcvConvArgs)).contents)] += 1
        return orig["vcv_conv_gemm"](ap, st)

    def dma(ap, pack, scratch, flip, valid, st):
        stat[conv_key("conv_dma_kernel", ctypes.cast(ap, ctypes.POINTER(_lib.VcvConvArgs)).contents)] += 1
        return orig["vcv_conv_dma_run"](ap, pack, scratch, flip, valid, st)

    def wgrad(ap, st):
        a = ctypes.cast(ap, ctypes.POINTER(_lib.VcvWgradArgs)).contents
        if L.vcv_conv_wgrad_takes_dma(ap):  # (the entry point hands these to wgrad_dma_kernel: not a fallback)
            return orig["vcv_conv_wgrad"](ap, st)
        stat[("conv_wgrad_kernel", "wgrad G=%d %d->%d K=%d s=%d dil=%d rows=%d x P=%d B=%d a_tf=%s b_tf=%s%s" % (
            a.G, a.Cg * a.G, a.Mg * a.G, a.K, a.s, abs(a.dj), a.Ta, a.P, a.B, TF.get(a.a_tf, a.a_tf), TF.get(a.b_tf, a.b_tf),
            " transposed" if a.transpose_out else ""))] += 1
        return orig["vcv_conv_wgrad"](ap, st)

    L.vcv_conv_gemm, L.vcv_conv_wgrad, L.vcv_conv_dma_run = gemm, wgrad, dma
    ops._FAMILIES.clear()  # (the launch wrapper caches the family tuples with the bound entry points: rebuilt with the wrappers)
    before = dict(ops.LAUNCH_COUNTS)
    try:
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
    finally:
        for n, f in orig.items():
            setattr(L, n, f)
        ops._FAMILIES.clear()
    total = {k: (ops.LAUNCH_COUNTS[k] - before.get(k, 0)) / steps for k in ops.LAUNCH_COUNTS if ops.LAUNCH_COUNTS[k] != before.get(k, 0)}
    return stat, total


def main():
    dev = torch.device("cuda:0")
    graphed.set_enabled(False)  # (the eager loop issues the calls; a replay issues none)
    out = ["# tools/fallback_census.py: launches per step that land on the three general-purpose kernels, by shape",
           "# (families: x3 = split-operand fp32, pk = packed fp32-input MFMA, bf16 / bf16io = bf16 operands / 16-bit activations,",
           "#  wgrad_x3 / wgrad_bf16 / wgrad = weight gradients; dma = conv_dma_kernel, gemm = conv_gemm_kernel, wgrad = wgrad_dma or conv_wgrad)"]
    cases = [("configs[1]: base vocoder fp32 B=16", "base", "vocoder", "f32", 16),
             ("configs[2]: base full bf16 B=32", "base", "full", "bf16", 32),
             ("configs[3] one rank: 48k full bf16 B=16", "48k", "full", "bf16", 16),
             ("configs[4]: 48k infer bf16 64 x 938 frames", "48k", "infer", "bf16", 64)]
    for title, config, workload, dtype, B in cases:
        cfg = configs.base() if config == "base" else configs.base_48k()
        ops.set_compute_dtype(dtype)
        torch.manual_seed(1234)
        if workload == "infer":
            sys.path.insert(0, ROOT)
            import bench
            run = bench.build_infer(cfg, B, 938, dev)
            warm, steps = 1, 1
        else:
            module = (VocoderGAN if workload == "vocoder" else VCVITS)(**cfg).to(dev)
            module.train()
            module.configure_optimizers()
            m = cfg["model"]
            make = synthetic.vocoder_batch if workload == "vocoder" else synthetic.full_batch
            batches = [make(B, m["inter_channels"] if workload == "vocoder" else m["hubert_channels"], seed=1234 + i, device=dev)
                       for i in range(2)]
            it = [0]

            def run():
                module.fit_batch(batches[it[0] % 2])
                it[0] += 1
            warm, steps = 3, 2
        for _ in range(warm):
            run()
        torch.cuda.synchronize()
        stat, total = census(run, steps)
        out.append("")
        out.append("## %s" % title)
        out.append("launches per step by family: " + ", ".join("%s %.0f" % kv for kv in sorted(total.items())))
        if not stat:
            out.append("(none on the general-purpose kernels)")
        for (kern, desc), n in sorted(stat.items(), key=lambda kv: (kv[0][0], -kv[1])):
            out.append("%6.1f  %-18s %s" % (n / steps, kern, desc))
        if workload != "infer":
            module.optim_g.close()
            module.optim_d.close()
            del module
        ops.invalidate_weights()
        ops.set_compute_dtype("f32")
        torch.cuda.empty_cache()
    text = "\n".join(out) + "\n"
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", "r6_fallback_census.txt"), "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
