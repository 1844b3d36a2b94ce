"""How many points each screening pass of u2_kmeans_assign_shadow leaves undecided on the two bench data sets (N = 1 M x 768, K = 300), and
what the E step costs with the coarse pass on and off.  usage: python tools/exp/km_undecided.py"""
import torch

from u2seg_amd.cluster import kmeans as KM

N, D, K = 1_000_000, 768, 300
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
centers = torch.randn((K, D), generator=g, device=dev) * 2
sets = {"mixture": (centers[torch.randint(0, K, (N,), generator=g, device=dev)] + 0.5 * torch.randn((N, D), generator=g, device=dev),
                    centers + 0.3 * torch.randn((K, D), generator=g, device=dev))}
xr = torch.randn((N, D), generator=g, device=dev)
sets["randn"] = (xr, xr[torch.randperm(N, generator=g, device=dev)[:K]].clone())


def timed(x, c, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        KM.assign(x, c)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, (x, c) in sets.items():
    KM._ws_cache.pop("assign:" + str(x.device), None)
    KM.assign(x, c)
    und, chk = KM.last_coarse_undecided(x.device), KM.last_recheck_count(x.device)
    for _ in range(3):
        KM.assign(x, c)
    print(name, "first call: coarse pass left %d undecided (%.1f %%), exact kernel re-checked %d;  steady state: coarse left %d, E step %.4f ms"
          % (und, 100.0 * und / N, chk, KM.last_coarse_undecided(x.device), timed(x, c)))
