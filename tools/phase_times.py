"""Debug aid: host time vs GPU-event time of the phases of one train step (finds host-bound / sync-bound phases)."""
import os, sys, time, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from u2seg_amd.config import get_cfg
from u2seg_amd.data import make_synthetic_batch
from u2seg_amd.modeling import build_model
from u2seg_amd.solver import build_optimizer

cfg = get_cfg(); cfg.merge_from_file(os.path.join(ROOT, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
torch.manual_seed(0)
model = build_model(cfg).cuda().train(); opt = build_optimizer(cfg, model)
batch = make_synthetic_batch(16, height=800, width=1333, device="cuda")
rec = collections.OrderedDict()
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        out = fn(*a, **k)
        e1.record(); t1 = time.perf_counter()
        rec.setdefault(label, []).append((t1 - t0, e0, e1))
        return out
    setattr(obj, name, w)
wrap(model, "_backbone_features", "backbone+fpn fwd")
wrap(model.sem_seg_head, "forward", "semseg fwd")
wrap(model.proposal_generator, "forward", "rpn fwd (all)")
wrap(model.proposal_generator, "predict_proposals", "  rpn predict_proposals")
wrap(model.proposal_generator, "label_and_sample_anchors", "  rpn label/sample")
wrap(model.roi_heads, "label_and_sample_proposals", "roi label/sample")
wrap(model.roi_heads, "_forward_box", "roi box fwd")
wrap(model.roi_heads, "_forward_mask", "roi mask fwd")
class Bwd: pass
def step():
    losses = model(batch)
    tot = sum(losses.values())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); tot.backward(); e1.record(); t1 = time.perf_counter()
    rec.setdefault("backward", []).append((t1 - t0, e0, e1))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); opt.step(); opt.zero_grad(); e1.record(); t1 = time.perf_counter()
    rec.setdefault("optimizer", []).append((t1 - t0, e0, e1))
for _ in range(3): step()
torch.cuda.synchronize(); rec.clear()
t0 = time.perf_counter()
for _ in range(4): step()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("wall ms/step %.2f" % ((t1 - t0) / 4 * 1e3))
for k, v in rec.items():
    n = len(v) / 4
    print("%-28s host %7.2f ms   gpu-span %7.2f ms   (calls/step %.0f)" % (k, sum(x[0] for x in v) / 4 * 1e3, sum(x[1].elapsed_time(x[2]) for x in v) / 4, n))
