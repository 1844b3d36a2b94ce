"""Debug aid: time u2_semseg_upsample_ce on the bench shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2seg_amd.layers import functional as F
torch.manual_seed(0)
b, h, w = 16, 200, 336
logits = torch.zeros((b, h, w, 32), dtype=torch.bfloat16, device="cuda")
logits[..., :28] = torch.randn((b, h, w, 28), device="cuda")
tgt = torch.randint(0, 28, (b, 4 * h, 4 * w), device="cuda").to(torch.uint8)
tgt[:, :, -44:] = 255
for _ in range(3):
    l = F.sem_seg_loss(logits.requires_grad_(), tgt, 28)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    l = F.sem_seg_loss(logits, tgt, 28)
e1.record(); torch.cuda.synchronize()
print("semseg fwd: %.3f ms, loss %.6f" % (e0.elapsed_time(e1) / 10, float(l)))
