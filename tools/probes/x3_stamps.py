"""Workgroup phase timing of conv_x3_kernel from in-kernel 100 MHz stamps (diagnostic build of conv_x3.hip with
-DVCV_X3_STAMPS linked into scratch/libvcvits_x3stamps.so; run with VCVITS_HIP_LIB pointing at it)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vcvits_amd import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
raw = ctypes.CDLL(os.environ["VCVITS_HIP_LIB"])
# name, B, C, M, T(or H), P, K, stride, pad, dil
CASES = [("gen c256 k3", 16, 256, 256, 256, 1, 3, 1, 1, 1), ("gen c256 k11", 16, 256, 256, 256, 1, 11, 1, 5, 1),
         ("gen c128 k3", 16, 128, 128, 2048, 1, 3, 1, 1, 1), ("gen c128 k7", 16, 128, 128, 2048, 1, 7, 1, 3, 1),
         ("gen c128 k11", 16, 128, 128, 2048, 1, 11, 1, 5, 1), ("gen c64 k7", 16, 64, 64, 8192, 1, 7, 1, 3, 1),
         ("gen c64 k11", 16, 64, 64, 8192, 1, 11, 1, 5, 1),
         ("discP2 conv4", 32, 1024, 1024, 102, 2, 5, 1, 2, 1), ("discP37 conv4", 32, 1024, 1024, 6, 37, 5, 1, 2, 1),
         ("discP2 conv3", 32, 512, 1024, 304, 2, 5, 3, 2, 1), ("discP2 conv2", 32, 128, 512, 911, 2, 5, 3, 2, 1)]
st = torch.zeros(1 << 20, device=dev, dtype=torch.int64)
for name, B, C, M, T, P, K, s, pd, d in CASES:
    x = torch.randn((B, C, T) if P == 1 else (B, C, T, P), device=dev)
    w = torch.randn(M, C, K, device=dev) * 0.05
    bias = torch.randn(M, device=dev)
    for _ in range(3):
        y = ops.conv_forward(x, w, bias, stride=s, pad=pd, dil=d)
    torch.cuda.synchronize()
    st.zero_()
    raw.vcv_x3_set_stamps(ctypes.c_void_p(st.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = ops.conv_forward(x, w, bias, stride=s, pad=pd, dil=d)
    e1.record()
    torch.cuda.synchronize()
    raw.vcv_x3_set_stamps(ctypes.c_void_p(0))
    sv = st.view(-1, 8).cpu()
    sv = sv[sv[:, 3] > 0]
    if sv.shape[0] == 0:
        print("%-14s (not on the split kernel)" % name); continue
    t0 = sv[:, 0].min()
    us = lambda a: float(a) * 0.01
    pro = (sv[:, 1] - sv[:, 0]).float() * 0.01
    main = (sv[:, 2] - sv[:, 1]).float() * 0.01
    epi = (sv[:, 3] - sv[:, 2]).float() * 0.01
    start = (sv[:, 0] - t0).float() * 0.01
    end = (sv[:, 3] - t0).float() * 0.01
    print("%-14s WGs %4d | span %6.1f us (event %6.1f) | start spread med %5.1f max %5.1f | prologue %5.2f  main %6.2f  epilogue %5.2f (medians; max %5.2f %6.2f %5.2f) | WG end med %6.1f" % (
        name, sv.shape[0], float(end.max()), e0.elapsed_time(e1) * 1e3, float(start.median()), float(start.max()),
        float(pro.median()), float(main.median()), float(epi.median()), float(pro.max()), float(main.max()), float(epi.max()), float(end.median())))
