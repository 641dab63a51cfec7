"""Phase timing of rel_attn_bwd_rows2_kernel from in-kernel 100 MHz stamps (diagnostic build, see attn_stamps.py).  The
forward of the training call runs first with the stamp buffer unset, then the backward alone stamps."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vcvits_amd import ops, _lib
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
ops.set_compute_dtype(dt)
dev = torch.device("cuda:0")
L = _lib.lib()
raw = ctypes.CDLL(os.environ["VCVITS_HIP_LIB"])
for B, H, dk, T in ((32, 4, 64, 204),):
    q, k, v, gy = (torch.randn(B, H * dk, T, device=dev).requires_grad_(True) for _ in range(4))
    ek, ev = torch.randn(1, 9, dk, device=dev).requires_grad_(True), torch.randn(1, 9, dk, device=dev).requires_grad_(True)
    mask = torch.ones(B, T, device=dev)
    nblk = ((T + 31) // 32 + 3) // 4
    st = torch.zeros(B * H * nblk * 4 * 8, device=dev, dtype=torch.int64)
    for rep in range(4):
        raw.vcv_attn_set_stamps(ctypes.c_void_p(0))
        o, _ = ops.rel_attention(q, k, v, ek, ev, mask, H, 4, 0.1, training=True, want_attn=False)
        torch.cuda.synchronize()
        st.zero_()
        raw.vcv_attn_set_stamps(ctypes.c_void_p(st.data_ptr()))
        o.backward(gy)
        torch.cuda.synchronize()
    raw.vcv_attn_set_stamps(ctypes.c_void_p(0))
    s = st.view(-1, 8).cpu()
    s = s[s[:, 7] > 0]
    t0 = s[:, 0].min()
    d = (s[:, 1:] - s[:, :-1]).float() * 0.01
    names = ["loads+R'+V stage", "dPd MFMAs", "barrier+K stage", "tile pass (P, dS)", "dQ MFMAs", "table grads", "-"]
    print("B=%d dtype=%s waves=%d  kernel span %.1f us" % (B, dt, s.shape[0], float(s[:, 7].max() - t0) * 0.01))
    for n, col in zip(names, d.t()):
        print("  %-20s median %6.2f us   max %6.2f" % (n, float(col.median()), float(col.max())))
