"""Timing of the level-2 ROIAlign gather (u2_roi_align_bwd_gather_sum) on synthetic ROI sets: where does a launch's time go?
   python tools/exp/roi_gather_bench.py            (repo root, on the GPU box)
Cases: no ROIs at all (the kernel's fixed cost + the epilogue), ROIs of 32 / 64 / 100 px at level 0 in 3 sets of 8192 + a mask set of 260,
with and without the two addend maps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import torch
from u2seg_amd.layers import functional as F

dev = "cuda:0"
from u2seg_amd import _hip
try:
    TRACE = ctypes.CDLL(_hip.lib_path()).u2_debug_gs_trace
except AttributeError:
    TRACE = None
B, H, W, C = 16, 200, 336, 256
shapes = [(B, H, W, C), (B, H // 2, W // 2, C), (B, H // 4, W // 4, C), (B, H // 8, W // 8, C)]
scales = [0.25, 0.125, 0.0625, 0.03125]
g = torch.Generator().manual_seed(0)


def make_set(n_per_img, size, P, frac_level0=1.0):
    n = n_per_img * B
    img = torch.arange(B).repeat_interleave(n_per_img).float()
    cx = torch.rand(n, generator=g) * 1344
    cy = torch.rand(n, generator=g) * 800
    s = size * (0.5 + torch.rand(n, generator=g))
    rois = torch.stack([img, (cx - s / 2).clamp(0, 1343), (cy - s / 2).clamp(0, 799), (cx + s / 2).clamp(1, 1344), (cy + s / 2).clamp(1, 800)], 1)
    lev = (torch.rand(n, generator=g) >= frac_level0).int() * 1
    rois, lev = rois.to(dev), lev.to(dev).int()
    order, seg = F._roi_group(rois, lev, B, 4)
    dout = torch.randn((n, P, P, C), generator=g).bfloat16().to(dev)
    return (rois, order, seg, dout, P, 1.0)


def timeit(sets, addends, reps=10):
    for _ in range(2):
        F._roi_gather(shapes, scales, sets, dev, level=0, addends=addends)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        F._roi_gather(shapes, scales, sets, dev, level=0, addends=addends)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


adds = (torch.randn(shapes[0], generator=g).bfloat16().to(dev), torch.randn(shapes[0], generator=g).bfloat16().to(dev))
for name, n_box, n_mask, size in (("no ROIs", 0, 0, 32), ("128 per image and set, 32 px", 128, 4, 32), ("128 per image and set, 64 px", 128, 4, 64),
                                  ("128, 100 px", 128, 4, 100), ("512 per image and set, 32 px", 512, 16, 32), ("512, 64 px", 512, 16, 64),
                                  ("512, 100 px", 512, 16, 100)):
    if n_box == 0:
        sets = [make_set(1, 32, 7, 0.0) for _ in range(3)] + [make_set(1, 32, 14, 0.0)]   # everything on level 1: level 0 sees no ROI
    else:
        sets = [make_set(n_box, size, 7) for _ in range(3)] + [make_set(n_mask, size, 14)]
    print("%-34s gather %.3f ms   + 2 addends %.3f ms" % (name, timeit(sets, ()), timeit(sets, adds)))
    if TRACE is not None:   # -DGS_TRACE build: shader-clock ticks of thread 0 per phase, averaged over the work-groups of one launch
        buf = (ctypes.c_ulonglong * 8)()
        TRACE(buf, 1)
        F._roi_gather(shapes, scales, sets, dev, level=0, addends=())
        torch.cuda.synchronize()
        TRACE(buf, 1)
        wg = max(buf[7], 1)
        print("    per work-group (ticks of 10 ns): scan %.0f  tables %.0f  staging %.0f  accumulation %.0f  loop rest %.0f  epilogue issue %.0f;  %.2f batches"
              % tuple([buf[k] / wg for k in range(6)] + [buf[6] / wg]))
