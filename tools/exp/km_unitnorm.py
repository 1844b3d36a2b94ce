"""The E step on L2-normalised rows with a common mean direction (what F.normalize(DINO features) looks like: usl-imagenet.py:103), N = 1 M x
768, K = 300: how many points each screening pass leaves, and the time.  usage: python tools/exp/km_unitnorm.py [mean_weight]"""
import sys

import torch
import torch.nn.functional as F

from u2seg_amd.cluster import kmeans as KM

N, D, K = 1_000_000, 768, 300
mw = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
m = F.normalize(torch.randn((1, D), generator=g, device=dev), dim=1)
cen = F.normalize(torch.randn((K, D), generator=g, device=dev), dim=1)
lab = torch.randint(0, K, (N,), generator=g, device=dev)
x = F.normalize(mw * m + 0.6 * cen[lab] + 0.6 * F.normalize(torch.randn((N, D), generator=g, device=dev), dim=1), dim=1)
c = x[torch.randperm(N, generator=g, device=dev)[:K]].clone()     # the reference's init: random rows
print("mean weight %.2f: |mean(x)| = %.3f, mean cosine between rows = %.3f" % (mw, float(x.mean(0).norm()), float((x[:2000] @ x[2000:4000].t()).mean())))
for it in range(4):
    KM._ws_cache.pop("assign:" + dev, None) if it == 0 else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    cl = KM.assign(x, c)
    e1.record()
    torch.cuda.synchronize()
    print("  iteration %d: E step %.3f ms, first pass left %s undecided, exact kernel re-checked %s" %
          (it, e0.elapsed_time(e1), KM.last_coarse_undecided(dev), KM.last_recheck_count(dev)))
    c, _ = KM.update(x, cl, K)
