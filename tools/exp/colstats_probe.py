"""u2_colstats (GroupNorm statistics) on the semantic head's shapes: per-image slots at batch 16 / 32.  usage: python tools/exp/colstats_probe.py"""
import torch

from u2seg_amd import _hip

dev = "cuda:0"
_hip.load()


def t(slots, rows, c, reps=20):
    x = torch.randn((slots * rows, c), device=dev).to(torch.bfloat16)
    out = torch.zeros((slots, 2, c), dtype=torch.float32, device=dev)
    for _ in range(3):
        _hip.call("u2_colstats", x, out, slots, rows, c, c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _hip.call("u2_colstats", x, out, slots, rows, c, c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("slots %3d rows %7d C %4d: %.4f ms  %.2f TB/s" % (slots, rows, c, ms, slots * rows * c * 2 / ms / 1e9))


for slots in (16, 32):
    for rows, c in ((200 * 336, 128), (100 * 168, 128), (50 * 84, 128), (25 * 42, 128), (200 * 336, 256)):
        t(slots, rows, c)
t(1, 32 * 200 * 336, 128)
