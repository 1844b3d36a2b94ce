"""Streaming rate of the normalisation passes on the step's big tensors (U2_EW_UNROLL variants are separate processes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from u2seg_amd import _hip
dev = "cuda"
BF = torch.bfloat16

def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
print("U2_EW_UNROLL =", os.environ.get("U2_EW_UNROLL", "2"))
for (rows, c) in [(16 * 200 * 336, 256), (16 * 100 * 168, 512), (16 * 200 * 336, 64), (16 * 50 * 84, 1024), (16 * 100 * 168, 128)]:
    x = torch.randn(rows, c, device=dev).to(BF); r = torch.randn(rows, c, device=dev).to(BF); out = torch.empty_like(x)
    dz = torch.empty_like(x); bits = torch.empty(rows * c // 8, dtype=torch.uint8, device=dev)
    sc = torch.rand(c, device=dev) + 0.5; sh = torch.randn(c, device=dev)
    mean = torch.zeros(c, device=dev); inv = torch.ones(c, device=dev); sums = torch.zeros(2 * c, device=dev)
    mb = rows * c * 2 / 1e6
    def aff_res(): _hip.call("u2_affine_act", x, sc, sh, r, out, 1, rows, c, c, 1, bits)
    def aff(): _hip.call("u2_affine_act", x, sc, sh, None, out, 1, rows, c, c, 1, None)
    def red3(): _hip.call("u2_norm_bwd_reduce", r, bits, x, mean, inv, sums, 1, rows, c, c, 1, None, None, None, dz, None, 1)
    def red2(): _hip.call("u2_norm_bwd_reduce", r, None, x, mean, inv, sums, 1, rows, c, c, 1, sc, sh, None, None, None, 0)
    def app0(): _hip.call("u2_norm_bwd_apply", dz, None, x, sc, sh, mean, out, None, 1, rows, c, c, 0, None, None)
    def app2(): _hip.call("u2_norm_bwd_apply", r, None, x, sc, sh, mean, out, None, 1, rows, c, c, 1, sc, sh)
    for name, fn, units in (("affine+res+relu+bits", aff_res, 3.06), ("affine+relu", aff, 2), ("reduce (bits, dz out)", red3, 3.06),
                            ("reduce (mask recomputed)", red2, 2), ("apply (no mask)", app0, 3), ("apply (mask recomputed)", app2, 3)):
        ms = timeit(fn)
        print("[%8d x %4d] %6.0f MB  %-26s %7.3f ms  %5.2f TB/s" % (rows, c, mb, name, ms, units * mb / ms / 1e6 * 1e3 / 1e3))
