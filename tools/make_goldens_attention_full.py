#!/usr/bin/env python3
"""Full-width attention fixtures from the REFERENCE's own MultiHeadAttention (run in the build container, where
/root/reference exists; the fixture travels, the reference does not).

  python3 tools/make_goldens_attention_full.py          -> tests/golden/attention_full.npz

The round-4 fused attention kernels (csrc/attention.hip, `vcv_rel_attn_fwd` / `vcv_rel_attn_bwd2`) take head widths
32 / 64 only, so the small fixture `attention.npz` (d_k = 8) never reaches them.  This one runs the reference module
(vits/model/transformer/relative_attention_transformer.py:103-251, window_size = 4, heads_share) at both configs' widths --
base: 256 channels, 4 heads (d_k = 64); 48k: 128 channels, 4 heads (d_k = 32) -- for T in {204, 256, 500} content frames
(204 = the synthetic utterance of SURVEY 8d, 256 = the r4 kernels' limit, 500 = an inference-length utterance) with
ragged lengths, forward and backward, and stores per case: the seeds that regenerate the weights and inputs
(golden_util.fill_state_dict / rng streams: numpy, machine-independent), and for the output, the attention
probabilities, the input gradient and every parameter gradient: (sum, abs-sum) in float64 and 256 sampled elements.
The oracle (oracle/vits_oracle.py: rel_attention) is asserted against the reference on every case at 1e-5 here, so a GPU
test that compares full tensors with the oracle and samples with this fixture is pinned to the reference twice over."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import make_goldens as MG  # noqa: E402  (installs the librosa / torchaudio / fairseq stubs and puts /root/reference on the path)
from golden_util import GOLDEN_DIR, checksum, rng_tensor  # noqa: E402
from oracle import vits_oracle as O  # noqa: E402

CASES = [(256, 4, 204), (256, 4, 256), (256, 4, 500), (128, 4, 204), (128, 4, 256), (128, 4, 500)]
B = 3
NS = 256


def lengths_for(T):
    return torch.tensor([T, (T * 2) // 3 + 1, T // 2 - 3])


def case_inputs(C, T, seed):
    rng = np.random.default_rng([seed, C, T])
    x = rng_tensor(rng, (B, C, T))
    r = rng_tensor(rng, (B, C, T))
    return x, r


def main():
    torch.set_num_threads(8)
    out = {"cases": np.array(CASES, dtype=np.int64), "batch": np.array(B)}
    for ci, (C, H, T) in enumerate(CASES):
        seed = 700 + ci
        mha = MG.MultiHeadAttention(C, C, H, p_dropout=0.0, window_size=4).eval()
        MG.load_seeded(mha, seed)
        x, r = case_inputs(C, T, seed)
        x.requires_grad_(True)
        la = lengths_for(T)
        xm = O.sequence_mask(la, T).unsqueeze(1).float()
        am = xm.unsqueeze(2) * xm.unsqueeze(-1)
        y = mha(x, x, attn_mask=am)
        params = list(mha.named_parameters())
        gr = torch.autograd.grad((y * r).sum(), [x] + [p for _, p in params], allow_unused=True)
        # the oracle against the reference, forward and backward, on this very case
        sdo = {"a." + k: v for k, v in mha.state_dict().items()}
        xo = x.detach().clone().requires_grad_(True)
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sdo.items()}
        yo, po = O.rel_attention(leaves, "a", xo, am, H, 4)
        go = torch.autograd.grad((yo * r).sum(), [xo] + [leaves["a." + n] for n, _ in params], allow_unused=True)
        MG.close(yo, y, what="attention out C=%d T=%d" % (C, T))
        MG.close(po, mha.attn, what="attention probs C=%d T=%d" % (C, T))
        for (n, _), a, b in zip([("x", None)] + params, go, gr):
            if b is None:
                continue
            MG.close(a, b, tol=2e-5, what="attention grad %s C=%d T=%d" % (n, C, T))
        tag = "c%d_" % ci
        out[tag + "seed"] = np.array(seed)
        out[tag + "lengths"] = la.numpy()
        for name, t in [("y", y), ("attn", mha.attn), ("dx", gr[0])] + [("dp_" + n, g) for (n, _), g in zip(params, gr[1:])]:
            if t is None:
                continue
            sums, idx, vals = checksum(t, n_samples=NS, seed=seed)
            out[tag + name + "_sums"], out[tag + name + "_idx"], out[tag + name + "_vals"] = sums, idx, vals
            out[tag + name + "_max"] = np.array(t.detach().abs().max().item())
        print("case %d: C=%d heads=%d T=%d lengths=%s  |y|max %.3f" % (ci, C, H, T, la.tolist(), y.abs().max().item()))
    path = os.path.join(GOLDEN_DIR, "attention_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
