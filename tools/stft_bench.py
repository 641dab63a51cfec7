"""Kernel-only timing of the STFT magnitude launches (forward / backward) at the training step's sizes:
python tools/stft_bench.py   -> us per launch and achieved GB/s over the algorithmic bytes (read 4 (T + 1536) B,
write 4 * 1025 * frames B; backward: dmag in, dy out, y re-read)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vcvits_amd import ops  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    dev = torch.device("cuda:0")
    print("launch                      frames      us    GB/s   (n_fft 2048, hop 512, zero pad 768)")
    for B, T in ((16, 16384), (32, 16384), (16, 196608), (32, 196608), (64, 480256)):
        y = (torch.rand(B, T, device=dev) * 1.8 - 0.9).requires_grad_(True)
        F = T // 512
        mag = ops.stft_mag(y)
        us = timed(lambda: ops.stft_mag(y.detach()))
        fb = 4.0 * B * (T + 1536) + 4.0 * B * 1025 * F
        print("fwd  B=%-3d T=%-7d %10d %8.1f %7.1f" % (B, T, B * F, us, fb / us / 1e3))
        if B * F <= 4096:
            d = torch.randn_like(mag)
            us = timed(lambda: torch.autograd.grad(ops.stft_mag(y), y, d))
            usf = timed(lambda: ops.stft_mag(y))
            bb = 4.0 * B * 1025 * F + 8.0 * B * T
            print("bwd  B=%-3d T=%-7d %10d %8.1f %7.1f   (fwd+bwd pair minus fwd)" % (B, T, B * F, us - usf, bb / max(us - usf, 1e-3) / 1e3))


if __name__ == "__main__":
    main()
