"""Debug aid: host vs GPU-event time of the phases of one panoptic inference batch."""
import os, sys, time, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from u2seg_amd.config import get_cfg
from u2seg_amd.data import make_synthetic_batch
from u2seg_amd.modeling import build_model
import u2seg_amd.modeling.panoptic_fpn as PF

cfg = get_cfg(); cfg.merge_from_file(os.path.join(ROOT, "configs/COCO-PanopticSegmentation/u2seg_eval_800.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", "cuda:0"])
torch.manual_seed(0)
model = build_model(cfg).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch = make_synthetic_batch(B, device="cuda:0")
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
wrap(model, "_backbone_features", "backbone+fpn")
wrap(model.sem_seg_head, "forward", "semseg head")
wrap(model.proposal_generator, "forward", "rpn")
wrap(model.roi_heads, "_forward_box", "roi box (+fast_rcnn_inference)")
wrap(model.roi_heads, "forward_with_given_boxes", "roi mask")
wrap(PF, "sem_seg_postprocess", "sem_seg_postprocess")
wrap(PF, "detector_postprocess", "detector_postprocess (mask paste)")
wrap(PF, "combine_semantic_and_instance_outputs", "panoptic merge")
with torch.no_grad():
    for _ in range(2): model(batch)
    torch.cuda.synchronize(); rec.clear()
    t0 = time.perf_counter()
    for _ in range(3): model(batch)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("wall ms/batch %.2f  (%.1f img/s)" % ((t1 - t0) / 3 * 1e3, B * 3 / (t1 - t0)))
for k, v in rec.items():
    print("%-36s host %8.2f ms   gpu-span %8.2f ms   (calls/batch %.0f)" % (k, sum(x[0] for x in v) / 3 * 1e3, sum(x[1].elapsed_time(x[2]) for x in v) / 3, len(v) / 3))
