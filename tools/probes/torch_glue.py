"""Which torch (ATen) device kernels a training step still launches beside the library's own: one step of a bench
workload under a TorchDispatchMode that counts every ATen op with a CUDA tensor argument by (op, shape, dtype) and by the
nearest vcvits_amd frame of the Python stack ('<autograd engine>' for the accumulation adds of the backward pass).
  python3 tools/probes/torch_glue.py [--workload full|vocoder] [--config base|48k] [--dtype bf16|f32] [--batch B]"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
from torch.utils._pytree import tree_flatten  # noqa: E402

SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.as_strided", "aten.t.", "aten.transpose",
        "aten.unsqueeze", "aten.squeeze", "aten.slice", "aten.select", "aten.expand", "aten.permute", "aten.reshape",
        "aten.empty", "aten.new_empty", "aten.split", "aten.unbind", "aten.narrow", "aten.lift_fresh", "aten._local_scalar",
        "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.result_type", "aten.chunk", "aten.flatten",
        "aten.unflatten", "aten.empty_like", "aten.empty_strided", "aten.set_", "aten.record_stream", "aten.is_pinned",
        "aten._pin_memory", "aten.resize_")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            flat, _ = tree_flatten((args, kwargs or {}))
            ts = [t for t in flat if isinstance(t, torch.Tensor) and t.is_cuda]
            if ts:
                big = max(ts, key=lambda t: t.numel())
                where = "<autograd engine>"
                for fr in reversed(traceback.extract_stack(limit=24)):
                    if "vcvits_amd" in fr.filename and "torch_glue" not in fr.filename:
                        where = "%s:%d" % (os.path.relpath(fr.filename, ROOT), fr.lineno)
                        break
                self.n[(name, tuple(big.shape), str(big.dtype).replace("torch.", ""), big.is_contiguous(), where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="full")
    ap.add_argument("--config", default="base")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=None)
    a = ap.parse_args()
    from vcvits_amd import configs, ops, synthetic
    from vcvits_amd.light import graphed
    from vcvits_amd.light.vcvits import VCVITS, VocoderGAN
    graphed.set_enabled(False)  # the eager loop is what issues the calls (a graph replay issues none: it replays them)
    dev = torch.device("cuda:0")
    ops.set_compute_dtype(a.dtype)
    cfg = configs.base() if a.config == "base" else configs.base_48k()
    B = a.batch or (32 if a.workload == "full" and a.config == "base" else 16)
    m = cfg["model"]
    torch.manual_seed(1234)
    module = (VocoderGAN if a.workload == "vocoder" else VCVITS)(**cfg).to(dev)
    module.train()
    module.configure_optimizers()
    make = synthetic.vocoder_batch if a.workload == "vocoder" else synthetic.full_batch
    width = m["inter_channels"] if a.workload == "vocoder" else m["hubert_channels"]
    batches = [make(B, width, seed=1234 + i, device=dev) for i in range(2)]
    for i in range(4):
        module.fit_batch(batches[i % 2])
    torch.cuda.synchronize()
    c = Count()
    with c:
        module.fit_batch(batches[0])
    torch.cuda.synchronize()
    tot = sum(c.n.values())
    print("# %s/%s/%s B=%d: %d ATen calls with a device tensor in one step" % (a.config, a.workload, a.dtype, B, tot))
    by_op = collections.Counter()
    for (name, shape, dt, contig, where), k in c.n.items():
        by_op[name] += k
    print("# by op: " + ", ".join("%s %d" % (n.replace("aten.", ""), k) for n, k in by_op.most_common(16)))
    print("%-26s %-30s %-8s %-6s %5s  %s" % ("op", "largest tensor", "dtype", "contig", "calls", "issued from"))
    for (name, shape, dt, contig, where), k in sorted(c.n.items(), key=lambda kv: -kv[1] * max(1, _numel(kv[0][1])))[:70]:
        print("%-26s %-30s %-8s %-6s %5d  %s" % (name.replace("aten.", ""), "x".join(map(str, shape)), dt, contig, k, where))


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


if __name__ == "__main__":
    main()
