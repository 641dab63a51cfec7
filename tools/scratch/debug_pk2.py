import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from vcvits_amd import ops
gpu = torch.device("cuda:0")
B, C, M, T, K, pad = 2, 256, 256, 384, 5, 2
x = torch.arange(B * C * T, dtype=torch.float32).reshape(B, C, T) % 1000 + 1
for k in range(K):
    w = torch.zeros(M, C, K)
    for m in range(M): w[m, m, k] = 1.0
    with torch.no_grad():
        yg = ops.conv1d(x.to(gpu), w.to(gpu), None, stride=1, pad=pad).cpu()
    yr = F.conv1d(x, w, None, padding=pad)
    bad = (yg != yr).nonzero()
    print("tap", k, "mismatches", bad.shape[0], "first", bad[:6].tolist())
    if bad.shape[0]:
        b, m, t = bad[0].tolist()
        print("   got", yg[b, m, max(t-2,0):t + 6].tolist(), "want", yr[b, m, max(t-2,0):t + 6].tolist())
