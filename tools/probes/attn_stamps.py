"""Phase timing of rel_attn_fwd_rows_kernel from in-kernel 100 MHz stamps (diagnostic build of attention.hip with
-DVCV_ATTN_STAMPS linked into scratch/libvcvits_stamps.so; run with VCVITS_HIP_LIB pointing at it)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vcvits_amd import ops, _lib
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
ops.set_compute_dtype(dt)
dev = torch.device("cuda:0")
L = _lib.lib()
raw = ctypes.CDLL(os.environ["VCVITS_HIP_LIB"])
for B, H, dk, T in ((32, 4, 64, 204), (16, 4, 64, 204)):
    q, k, v = (torch.randn(B, H * dk, T, device=dev) for _ in range(3))
    ek, ev = torch.randn(1, 9, dk, device=dev), torch.randn(1, 9, dk, device=dev)
    mask = torch.ones(B, T, device=dev)
    nblk = ((T + 31) // 32 + 3) // 4
    st = torch.zeros(B * H * nblk * 4 * 8, device=dev, dtype=torch.int64)
    raw.vcv_attn_set_stamps(ctypes.c_void_p(st.data_ptr()))
    with torch.no_grad():
        for _ in range(5):
            ops.rel_attention(q, k, v, ek, ev, mask, H, 4, want_attn=False)
        torch.cuda.synchronize()
        st.zero_()
        ops.rel_attention(q, k, v, ek, ev, mask, H, 4, want_attn=False)
    torch.cuda.synchronize()
    s = st.view(-1, 8).cpu()
    s = s[s[:, 7] > 0]  # waves that ran to the end
    t0 = s[:, 0].min()
    d = (s[:, 1:] - s[:, :-1]).float() * 0.01  # us
    names = ["staging+barrier", "mask+R", "QK", "scale/mask/exp", "P/dropout", "band+PV", "store"]
    print("B=%d dtype=%s waves=%d  kernel span %.1f us; first start..last start %.1f us" % (
        B, dt, s.shape[0], float(s[:, 7].max() - t0) * 0.01, float(s[:, 0].max() - t0) * 0.01))
    for n, col in zip(names, d.t()):
        print("  %-18s median %6.2f us   max %6.2f" % (n, float(col.median()), float(col.max())))
    print("  per-wave total     median %6.2f us" % float(((s[:, 7] - s[:, 0]).float() * 0.01).median()))
