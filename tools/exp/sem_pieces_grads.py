"""Gradient arena after one backward pass with the semantic head launched in pieces / as a whole (same weights, batch and seeds):
per-parameter relative L2 difference, mode against mode and mode against itself (the run-to-run noise).  python tools/exp/sem_pieces_grads.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from u2seg_amd.config import get_cfg  # noqa: E402
from u2seg_amd.data import make_synthetic_batch  # noqa: E402
from u2seg_amd.layers import functional as F  # noqa: E402
from u2seg_amd.modeling import build_model  # noqa: E402
from u2seg_amd.solver import build_optimizer  # noqa: E402

dev = "cuda:0"
batch = make_synthetic_batch(16, start_index=0, height=800, width=1333, device=dev)
torch.manual_seed(1234)
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", dev, "SOLVER.IMS_PER_BATCH", 16])
model = build_model(cfg)
model.train()
opt = build_optimizer(cfg, model)
names = {id(p): n for n, p in model.named_parameters()}


def grads(mode):
    os.environ["U2_SEM_PIECES"] = mode
    torch.manual_seed(1000)
    opt.zero_grad()
    ld = model(batch)
    sum(ld.values()).backward()
    F.assert_no_deferred_gradients()
    F.join_all_streams()
    torch.cuda.synchronize()
    return opt.flat_grad.clone(), {k: float(v) for k, v in ld.items()}


grads("0")  # warm-up (scratch allocations)
runs = {m: grads(m[0]) for m in ("1a", "0a", "1b", "0b")}


def compare(a, b):
    ga, gb = runs[a][0], runs[b][0]
    worst = []
    for p, off in zip(opt.params, opt.param_offset):
        n = p.numel()
        x, y = ga[off:off + n], gb[off:off + n]
        den = float(y.norm())
        worst.append((float((x - y).norm()) / max(den, 1e-30), names[id(p)], den))
    worst.sort(reverse=True)
    tot = float((ga - gb).norm()) / float(gb.norm())
    print("%s vs %s: whole arena rel L2 %.3e; worst parameters: " % (a, b, tot) + ", ".join("%s %.2e (|g| %.2e)" % (n, r, d) for r, n, d in worst[:6]))
    sem = [(r, n) for r, n, d in worst if n.startswith("sem_seg_head")]
    print("   semantic head: max %.2e (%s), median %.2e" % (sem[0][0], sem[0][1], sem[len(sem) // 2][0]))


compare("1a", "1b")
compare("0a", "0b")
compare("1a", "0a")
compare("1b", "0b")
