import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from vcvits_amd import ops
gpu = torch.device("cuda:0")
for case in [(2, 32, 128, 67, 3, 5, 3, 2), (1, 32, 128, 67, 3, 5, 3, 2), (1, 32, 128, 190, 1, 5, 3, 2), (1, 32, 128, 400, 2, 5, 3, 2)]:
    B, C, M, H, P, K, s, p = case
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, P, generator=gen)
    w = torch.randn(M, C, K, 1, generator=gen) / (C * K) ** 0.5
    xr, wr = (t.clone().requires_grad_(True) for t in (x, w))
    yr = F.conv2d(xr, wr, None, stride=(s, 1), padding=(p, 0))
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xg, wg = (t.clone().to(gpu).requires_grad_(True) for t in (x, w))
    yg = ops.conv1d(xg, wg, None, stride=s, pad=p)
    yg.backward(gy.to(gpu))
    d = (wg.grad.cpu() - wr.grad)[..., 0]
    ref = wr.grad[..., 0]
    print(case, "U", yr.shape[2] * P, "rel", float(d.norm() / ref.norm()))
    print("  per tap:", [round(float(d[:, :, k].norm() / ref[:, :, k].norm()), 4) for k in range(K)])
    print("  per c (first 8):", [round(float(d[:, c].norm() / ref[:, c].norm()), 3) for c in range(8)], " last:", [round(float(d[:, c].norm() / ref[:, c].norm()), 3) for c in range(C - 4, C)])
    print("  per m blocks:", [round(float(d[m:m + 32].norm() / ref[m:m + 32].norm()), 3) for m in range(0, M, 32)])
    # contribution test: only positions in the first chunk
print("stride 1 cases")
for case in [(1, 32, 128, 64, 1, 5, 1, 2), (2, 32, 128, 100, 3, 5, 1, 2)]:
    B, C, M, H, P, K, s, p = case
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, P, generator=gen)
    w = torch.randn(M, C, K, 1, generator=gen) / (C * K) ** 0.5
    xr, wr = (t.clone().requires_grad_(True) for t in (x, w))
    yr = F.conv2d(xr, wr, None, stride=(s, 1), padding=(p, 0))
    gy = torch.randn(yr.shape, generator=gen)
    yr.backward(gy)
    xg, wg = (t.clone().to(gpu).requires_grad_(True) for t in (x, w))
    yg = ops.conv1d(xg, wg, None, stride=s, pad=p)
    yg.backward(gy.to(gpu))
    d = (wg.grad.cpu() - wr.grad)[..., 0]
    ref = wr.grad[..., 0]
    print(case, "U", yr.shape[2] * P, "rel", float(d.norm() / ref.norm()))
