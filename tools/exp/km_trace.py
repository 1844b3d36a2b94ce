"""Where a work-group of the k-means coarse pass spends its time (build with -DU2_KM_TRACE): entry -> first barrier (ring fill), the
24-step loop, the arg-min epilogue; per role wave.  usage: python tools/exp/km_trace.py"""
import ctypes

import numpy as np
import torch

from u2seg_amd import _hip
from u2seg_amd.cluster import kmeans as KM

N, D, K = 1_000_000, 768, 300
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
centers = torch.randn((K, D), generator=g, device=dev) * 2
x = centers[torch.randint(0, K, (N,), generator=g, device=dev)] + 0.5 * torch.randn((N, D), generator=g, device=dev)
c = centers + 0.3 * torch.randn((K, D), generator=g, device=dev)
for _ in range(3):
    KM.assign(x, c)
torch.cuda.synchronize()
lib = _hip.load()
nwg = (N + 255) // 256
buf = np.zeros(nwg * 16, dtype=np.uint64)
lib.u2_km_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.u2_km_trace_dump(buf.ctypes.data, buf.size) == 0
t = buf.reshape(nwg, 2, 8).astype(np.int64)
t = t[t[:, 0, 0] != 0]          # persistent coarse pass: one work-group per CU, the other rows stay empty
for role, name in ((0, "wave 0 (centroid requests)"), (1, "wave 4 (x requests)")):
    d = np.diff(t[:, role, :4], axis=1)
    life = t[:, role, 4] - t[:, role, 0]
    print(name, "ticks: fill %.0f  loop of the first tile %.0f (%.0f per step)  its arg-min %.0f;  work-group life: median %.0f max %.0f (%d traced)" % (
        np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 1]) / (D // 32), np.median(d[:, 2]), np.median(life), life.max(), len(t)))
# the launch as a whole, in ticks of one XCD's counter: span between the earliest entry and the latest exit among work-groups whose
# stamps are close (same time base)
span = t[:, 0, 4].max() - t[:, 0, 0].min()
print("raw span (mixes time bases):", span)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    KM.assign(x, c)
e1.record()
torch.cuda.synchronize()
print("assign %.4f ms" % (e0.elapsed_time(e1) / 10))
