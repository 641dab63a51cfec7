import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from vcvits_amd import ops
from vcvits_amd._lib import ACT_LEAKY
dev = torch.device("cuda:0")
torch.manual_seed(0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 19
chans = [1, 32, 128, 512, 1024, 1024]
H = (16384 + P - 1) // P
for B in (2,):
  h = H
  for i in range(5):
    s = 3 if i < 4 else 1
    ci, co = chans[i], chans[i+1]
    x = torch.randn(B, ci, h, P)
    w = torch.randn(co, ci, 5) * (ci*5) ** -0.5
    b = torch.randn(co) * 0.1
    xc = x.clone().requires_grad_(True); wc = w.clone().requires_grad_(True)
    y = F.leaky_relu(F.conv2d(xc, wc.unsqueeze(-1), b, stride=(s,1), padding=(2,0)), 0.1)
    r = torch.randn_like(y)
    (y*r).sum().backward()
    xg = x.to(dev).requires_grad_(True); wg = w.to(dev).requires_grad_(True); bg = b.to(dev).requires_grad_(True)
    yg = ops.conv1d(xg, wg.unsqueeze(-1), bg, stride=s, pad=2, out_act=ACT_LEAKY, slope=0.1)
    (yg*r.to(dev)).sum().backward()
    e = lambda a, c: ((a.cpu()-c).abs().max() / c.abs().max()).item()
    print("P", P, "layer", i, "H", h, "U", y.shape[2]*P, "fwd", "%.2e" % e(yg.detach(), y.detach()), "dx", "%.2e" % e(xg.grad, xc.grad), "dw", "%.2e" % e(wg.grad, wc.grad))
    if e(xg.grad, xc.grad) > 1e-4:
        d = (xg.grad.cpu()-xc.grad).abs()
        idx = (d > 1e-4*xc.grad.abs().max()).nonzero()
        print("  bad elements", idx.shape[0], "rows(h):", sorted(set(idx[:,2].tolist()))[:40], "cols:", sorted(set(idx[:,3].tolist()))[:40], "ch:", sorted(set(idx[:,1].tolist()))[:10], "b:", sorted(set(idx[:,0].tolist())))
    h = y.shape[2]
