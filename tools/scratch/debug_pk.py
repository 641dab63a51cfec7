import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from vcvits_amd import ops
gpu = torch.device("cuda:0")
for case in [(2, 1024, 1024, 16, 13, 5, 1, 2), (2, 256, 512, 384, 1, 5, 1, 2)]:
    B, C, M, H, P, K, s, p = case
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, P, generator=gen)
    w = torch.randn(M, C, K, 1, generator=gen) / (C * K) ** 0.5
    yr = F.conv2d(x, w, None, stride=(s, 1), padding=(p, 0))
    xg, wg = x.to(gpu), w.to(gpu)
    with torch.no_grad():
        yg = ops.conv1d(xg, wg, None, stride=s, pad=p).cpu()
    d = (yg - yr)
    U = yr.shape[2] * yr.shape[3]
    d2 = d.reshape(B, M, U); r2 = yr.reshape(B, M, U)
    print(case, "rel", float(d.norm() / yr.norm()))
    print("  per m block of 32:", [round(float(d2[:, m:m + 32].norm() / r2[:, m:m + 32].norm()), 3) for m in range(0, min(M, 256), 32)])
    print("  per u block of 32:", [round(float(d2[:, :, u:u + 32].norm() / r2[:, :, u:u + 32].norm()), 3) for u in range(0, min(U, 384), 32)])
    print("  ratio yg/yr sample", yg.reshape(-1)[:4].tolist(), yr.reshape(-1)[:4].tolist())
