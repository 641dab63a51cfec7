"""Distance to the FLOAT64 oracle step of the HIP training step in each of the library's three fp32 arithmetics (six-term
split, nine-term split = exact operands, fp32-input MFMA = bit-for-bit an fmaf chain) and of the torch-CPU fp32 oracle step,
on the headline workload (base widths, generator + MPD + MSD) at B = 2 and B = 16.  Whole-gradient relative L2 per network.
Output: profiles/r6_f64_rank_arithmetics.txt          python3 tools/f64_rank.py   (GPU box; ~2 min of CPU float64)"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle.cpu_step import CpuTrainer  # noqa: E402
from vcvits_amd import configs, ops, synthetic  # noqa: E402
from vcvits_amd.light.vcvits import DEFAULT_PERIODS, VocoderGAN  # noqa: E402


def net_dist(grads, ref64):
    acc = {}
    for k, r in ref64.items():
        n = acc.setdefault(k.split(".")[0], [0.0, 0.0])
        n[0] += (grads[k].double() - r.double()).pow(2).sum().item()
        n[1] += r.double().pow(2).sum().item()
    return {k: (a / b) ** 0.5 for k, (a, b) in acc.items()}


def main():
    dev = torch.device("cuda:0")
    torch.set_num_threads(16)
    out = ["# tools/f64_rank.py: relative L2 distance of the WHOLE gradient of each network to the float64 oracle step",
           "# (one G step + one D step, base widths, HiFi-GAN generator + MPD + MSD, seed 7 weights, synthetic batch)",
           "%-4s %-44s %12s %14s %13s   %s" % ("B", "step computed by", "net_g", "net_period_d", "net_scale_d", "losses (g, d)")]
    for B in (2, 16):
        torch.manual_seed(7)
        cfg = configs.base()
        sd = copy.deepcopy(VocoderGAN(**cfg).state_dict())
        batch = synthetic.vocoder_batch(B, cfg["model"]["inter_channels"], seed=1234)
        t64 = CpuTrainer(copy.deepcopy(sd), cfg, DEFAULT_PERIODS, vocoder_only=True, dtype=torch.float64)
        l64 = t64.batch(batch)
        ref64 = dict(t64.grads_g)
        ref64.update(t64.grads_d)
        t32 = CpuTrainer(copy.deepcopy(sd), cfg, DEFAULT_PERIODS, vocoder_only=True)
        l32 = t32.batch(batch)
        g32 = dict(t32.grads_g)
        g32.update(t32.grads_d)
        rows = [("float64 oracle (the yardstick)", {k: 0.0 for k in ("net_g", "net_period_d", "net_scale_d")}, l64),
                ("torch-CPU fp32 oracle", net_dist(g32, ref64), l32)]
        for name, split in (("HIP, split operands, 6 product terms (default)", (True, 6)), ("HIP, split operands, 9 terms (exact operands)", (True, 9)),
                            ("HIP, fp32-input MFMA (fmaf chain)", (False, None))):
            ops.set_f32_split(split[0], terms=split[1], wgrad=split[0])
            module = VocoderGAN(**cfg)
            module.load_state_dict(sd)
            module = module.to(dev)
            module.configure_optimizers()
            names = {id(p): n for n, p in module.named_parameters()}
            grads = {}

            def probe(idx, opt):
                for p in opt.params:
                    grads[names[id(p)]] = p.grad.detach().cpu().clone()
            o = module.fit_batch({k: v.to(dev) for k, v in batch.items()}, after_backward=probe)
            rows.append((name, net_dist(grads, ref64), (o["g"], o["d"])))
            module.optim_g.close()
            module.optim_d.close()
            del module
            ops.invalidate_weights()
        ops.set_f32_split(True, terms=6, wgrad=True)
        for name, d, ls in rows:
            out.append("%-4d %-44s %12.3e %14.3e %13.3e   %.9g %.9g" % (B, name, d["net_g"], d["net_period_d"], d["net_scale_d"], float(ls[0]), float(ls[1])))
        out.append("")
    text = "\n".join(out) + "\n"
    open(os.path.join(ROOT, "profiles", "r6_f64_rank_arithmetics.txt"), "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
