"""Host cost per launcher call (tiny tensors: the GPU is never the bound): raw conv_forward, autograd conv1d forward,
forward + backward.  python tools/probes/host_call_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vcvits_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(2, 64, 256, device=dev)
w = torch.randn(64, 64, 3, device=dev, requires_grad=True)
b = torch.randn(64, device=dev, requires_grad=True)
xg = x.clone().requires_grad_(True)
N = 3000


def t(fn, n=N):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6


with torch.no_grad():
    print("conv_forward (raw launcher, no autograd): %6.1f us" % t(lambda: ops.conv_forward(x, w, b, pad=1)))
    print("conv1d under no_grad (autograd Function):  %6.1f us" % t(lambda: ops.conv1d(x, w, b, pad=1)))
print("conv1d forward with grad:                  %6.1f us" % t(lambda: ops.conv1d(xg, w, b, pad=1)))


def fb():
    y = ops.conv1d(xg, w, b, pad=1, out_act=ops.ACT_LEAKY)
    y.backward(y)


print("conv1d forward + backward (dgrad, wgrad, bias, act): %6.1f us" % t(fb, 1000))
print("torch.empty:                               %6.1f us" % t(lambda: torch.empty((2, 64, 256), device=dev)))
