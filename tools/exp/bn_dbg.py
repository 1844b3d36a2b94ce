import sys, torch
sys.path.insert(0,'/root/repo')
import torch.nn.functional as TF
from u2seg_amd import _hip
from u2seg_amd.layers import functional as F
_hip.load()
DEV='cuda:0'
def bf(t): return t.bfloat16().float()
def nhwc(t): return t.permute(0,2,3,1).contiguous().bfloat16().to(DEV)
def nchw(t): return t.permute(0,3,1,2).float().cpu()
g = torch.Generator().manual_seed(5)
x = bf(torch.randn((3, 64, 11, 13), generator=g) * 2 + 0.5)
res = bf(torch.randn((3, 64, 11, 13), generator=g))
gamma = (1 + 0.2 * torch.randn(64, generator=g)).requires_grad_(True)
beta = (0.1 * torch.randn(64, generator=g)).requires_grad_(True)
rm, rv = torch.zeros(64), torch.ones(64)
xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
pre = TF.batch_norm(xr, rm, rv, gamma, beta, True, 0.1, 1e-5)
yr = bf(TF.relu(bf(pre) + rr))
gy = bf(torch.randn(yr.shape, generator=g))
yr.backward(gy)
xd, rd = nhwc(x).requires_grad_(True), nhwc(res).requires_grad_(True)
gd, bd = gamma.detach().to(DEV).requires_grad_(True), beta.detach().to(DEV).requires_grad_(True)
rmd, rvd = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
xf = nchw(xd)
stats = torch.stack([xf.sum((0, 2, 3)), (xf * xf).sum((0, 2, 3))]).to(DEV)
y = F.batch_norm_act(xd, stats, gd, bd, rmd, rvd, rd, True, 0.1, 1e-5)
print('fwd maxdiff', float((nchw(y)-yr.detach()).abs().max()), 'mask mismatch', int(((nchw(y)>0)!=(yr.detach()>0)).sum()))
y.backward(nhwc(gy))
d=(nchw(xd.grad)-xr.grad).abs()
print('dx maxdiff', float(d.max()), 'ref max', float(xr.grad.abs().max()), 'n bad', int((d>0.05).sum()), 'of', d.numel())
dr=(nchw(rd.grad)-rr.grad).abs()
print('dres maxdiff', float(dr.max()), 'n bad', int((dr>0.05).sum()))
print('dgamma', float((gd.grad.cpu()-gamma.grad).abs().max()), 'dbeta', float((bd.grad.cpu()-beta.grad).abs().max()))
idx=(d>0.05).nonzero()[:5]
for i in idx:
    i=tuple(i.tolist()); print(i, float(nchw(xd.grad)[i]), float(xr.grad[i]), 'y', float(nchw(y)[i]), float(yr[i]), 'gy', float(gy[i]))
