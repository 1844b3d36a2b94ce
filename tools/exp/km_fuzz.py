"""Random shapes through the screened k-means E step (shadow path forced on): labels must equal the exact-fp32 kernel's.
usage: python tools/exp/km_fuzz.py [cases] [seed]"""
import sys

import torch

from u2seg_amd.cluster import kmeans as KM

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
KM.SHADOW_MIN_POINTS = 256
dev = "cuda:0"
g = torch.Generator().manual_seed(seed)
bad = 0
for i in range(cases):
    d = [32, 64, 96, 128, 256, 384, 768][int(torch.randint(0, 7, (1,), generator=g))]
    k = int(torch.randint(2, 1281, (1,), generator=g)) if i % 3 else int(torch.randint(2, 321, (1,), generator=g))
    n = int(torch.randint(256, 150000, (1,), generator=g))
    kind = i % 4
    if kind == 0:      # clustered
        cen = torch.randn((k, d), generator=g) * 2
        x = cen[torch.randint(0, k, (n,), generator=g)] + 0.5 * torch.randn((n, d), generator=g)
        c = cen + 0.3 * torch.randn((k, d), generator=g)
    elif kind == 1:    # unstructured, centroids = rows
        x = torch.randn((n, d), generator=g)
        c = x[torch.randperm(n, generator=g)[:k]].clone() if k <= n else torch.randn((k, d), generator=g)
    elif kind == 2:    # bf16-exact values, tiny and huge norms, duplicates
        x = (torch.randn((n, d), generator=g) * 3).bfloat16().float()
        x[: n // 10] *= 1e-3
        x[n // 10: n // 5] *= 1e3
        c = x[torch.randperm(n, generator=g)[:k]].clone() if k <= n else torch.randn((k, d), generator=g)
        if k > 3:
            c[k - 1] = c[0]
    else:              # offset data (large common mean)
        x = torch.randn((n, d), generator=g) + 10.0
        c = x[torch.randperm(n, generator=g)[:k]].clone() if k <= n else torch.randn((k, d), generator=g) + 10.0
    xd, cd = x.to(dev), c.to(dev)
    KM._ws_cache.pop("assign:" + dev, None)
    fast = KM.assign(xd, cd)
    und, chk = KM.last_coarse_undecided(dev), KM.last_recheck_count(dev)
    fast2 = KM.assign(xd, cd)          # second call: the screening state may have switched the first pass off
    exact = KM.assign(xd, cd, exact=True)
    ok = bool(torch.equal(fast, exact)) and bool(torch.equal(fast2, exact))
    bad += not ok
    print("%s case %2d kind %d N %6d D %3d K %4d: first pass left %s undecided, exact kernel re-checked %s%s"
          % ("ok " if ok else "BAD", i, kind, n, d, k, und, chk, "" if ok else "  MISMATCH %d" % int((fast != exact).sum())))
    del xd, cd
    KM.release_shadow()
print("%d of %d cases differ" % (bad, cases))
sys.exit(1 if bad else 0)
