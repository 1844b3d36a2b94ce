"""Autograd nodes whose output feeds two or more consumers (= gradient sums formed by the autograd engine with aten::add)."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from u2seg_amd.config import get_cfg
from u2seg_amd.data import make_synthetic_batch
from u2seg_amd.modeling import build_model

dev = "cuda"
torch.manual_seed(1234)
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", dev, "SOLVER.IMS_PER_BATCH", 16])
model = build_model(cfg); model.train()
batch = make_synthetic_batch(16, start_index=0, height=800, width=1333, device=dev)
losses = model(batch)
loss = sum(losses.values())
edges = collections.Counter()
consumers = collections.defaultdict(list)
seen = set()
stack = [loss.grad_fn]
while stack:
    fn = stack.pop()
    if fn is None or id(fn) in seen:
        continue
    seen.add(id(fn))
    for nxt, idx in fn.next_functions:
        if nxt is None:
            continue
        edges[(nxt, idx)] += 1
        consumers[(nxt, idx)].append(type(fn).__name__)
        stack.append(nxt)
rows = []
for (fn, idx), c in edges.items():
    if c < 2:
        continue
    try:
        md = fn._input_metadata[idx]
        shape, dt = tuple(md.shape), str(md.dtype).replace("torch.", "")
    except Exception:
        shape, dt = "?", "?"
    numel = 1
    if shape != "?":
        for d in shape: numel *= d
    rows.append((numel, type(fn).__name__, idx, c, shape, dt, consumers[(fn, idx)]))
rows.sort(key=lambda r: -r[0])
for numel, name, idx, c, shape, dt, cons in rows[:40]:
    print("%-28s out %d  x%d  %s %s  <- %s" % (name, idx, c, shape, dt, ", ".join(cons)))
