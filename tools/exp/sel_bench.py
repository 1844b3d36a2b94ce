"""Times u2_topk_rows on the shapes of the training step (batch 16, 800x1333)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from u2seg_amd.layers import functional as F

dev = "cuda"
torch.manual_seed(0)

def timeit(name, fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-60s %8.1f us" % (name, e0.elapsed_time(e1) / reps * 1e3), flush=True)

b = 16
A = 268569
key = torch.rand(b, A, device=dev)
labels = torch.full((b, A), -1, dtype=torch.int8, device=dev)
r = torch.rand(b, A, device=dev)
labels[r < 0.90] = 0
labels[r > 0.9995] = 1
timeit("rpn subsample pos: fp32 keys [16, 268569] mask==1 k=128", lambda: F.topk_rows(key, 128, largest=False, mask=labels, mask_value=1, want_vals=False))
timeit("rpn subsample neg: fp32 keys [16, 268569] mask==0 k=256", lambda: F.topk_rows(key, 256, largest=False, mask=labels, mask_value=0, want_vals=False))
for (h, w) in [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]:
    o = (torch.randn(b, h, w, 32, device=dev) * 2 - 4).bfloat16()
    hwa = h * w * 3
    k = min(hwa, 2000)
    timeit("rpn level top-k: bf16 [16, %d] group 3 pitch 32 k=%d" % (hwa, k), lambda: F.topk_rows(o, k, largest=True, group=3, pitch=32, n=hwa))
n = 2000 * 4 + 819
scores = torch.randn(b, n, device=dev).bfloat16().float()
keep = (torch.rand(b, n, device=dev) < 0.97).to(torch.int8)
timeit("rpn score order: fp32 [16, %d] full sort" % n, lambda: F.topk_rows(scores, n, largest=True, mask=keep, mask_value=1, want_vals=False))
m = 2000 + 20
key2 = torch.rand(b, m, device=dev)
kind = (torch.rand(b, m, device=dev) < 0.1).to(torch.int8) + 1
timeit("roi sample fg: fp32 [16, %d] k=128" % m, lambda: F.topk_rows(key2, 128, largest=False, mask=kind, mask_value=2, want_vals=False))
timeit("roi sample bg: fp32 [16, %d] k=512" % m, lambda: F.topk_rows(key2, 512, largest=False, mask=kind, mask_value=1, want_vals=False))
maps = [(torch.randn(b, h, w, 32, device=dev) * 2 - 4).bfloat16() for (h, w) in [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]]
timeit("rpn level top-k, all five levels in one multi call", lambda: F.topk_rows_multi(
    [dict(vals=o, k=min(o.shape[1] * o.shape[2] * 3, 2000), largest=True, group=3, pitch=32, n=o.shape[1] * o.shape[2] * 3) for o in maps]))
timeit("rpn subsample pos + neg in one multi call", lambda: F.topk_rows_multi([
    dict(vals=key, k=128, largest=False, mask=labels, mask_value=1, want_vals=False),
    dict(vals=key, k=256, largest=False, mask=labels, mask_value=0, want_vals=False)]))
timeit("roi sample fg + bg in one multi call", lambda: F.topk_rows_multi([
    dict(vals=key2, k=128, largest=False, mask=kind, mask_value=2, want_vals=False),
    dict(vals=key2, k=512, largest=False, mask=kind, mask_value=1, want_vals=False)]))
