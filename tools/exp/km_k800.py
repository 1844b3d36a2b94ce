"""k-means E / M step at K = 800 (the cluster count of u2seg_R50_800) on 1 M x 768 mixture data.  usage: python tools/exp/km_k800.py [K]"""
import sys

import torch

from u2seg_amd.cluster import kmeans as KM

N, D = 1_000_000, 768
K = int(sys.argv[1]) if len(sys.argv) > 1 else 800
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
centers = torch.randn((K, D), generator=g, device=dev) * 2
x = centers[torch.randint(0, K, (N,), generator=g, device=dev)] + 0.5 * torch.randn((N, D), generator=g, device=dev)
c = centers + 0.3 * torch.randn((K, D), generator=g, device=dev)


def timed(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


lab = KM.assign(x, c)
print("K = %d: assign %.3f ms, update %.3f ms, undecided after the first pass %s, re-checked exactly %s"
      % (K, timed(lambda: KM.assign(x, c)), timed(lambda: KM.update(x, lab, K)), KM.last_coarse_undecided(x.device), KM.last_recheck_count(x.device)))
ex = KM.assign(x[:50000].clone(), c, exact=True)
print("labels == exact kernel's on the first 50 000 rows:", bool(torch.equal(lab[:50000], ex)))
