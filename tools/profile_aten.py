"""Debug aid: attribute ATen glue kernels of one train step to python call sites (torch.profiler, with_stack)."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.profiler import profile, ProfilerActivity
from u2seg_amd.config import get_cfg
from u2seg_amd.data import make_synthetic_batch
from u2seg_amd.modeling import build_model
from u2seg_amd.solver import build_optimizer

cfg = get_cfg(); cfg.merge_from_file(os.path.join(ROOT, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
torch.manual_seed(0)
model = build_model(cfg).cuda().train(); opt = build_optimizer(cfg, model)
batch = make_synthetic_batch(16, height=800, width=1333, device="cuda")
def step():
    losses = model(batch); sum(losses.values()).backward(); opt.step(); opt.zero_grad()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
ev = prof.events()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in ev:
    if e.device_type.name != "CPU" or e.self_device_time_total <= 0: continue
    if not e.name.startswith("aten::"): continue
    site = "?"
    for fr in (e.stack or []):
        if "u2seg_amd" in fr or "bench" in fr: site = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr[-70:]; break
    if site == "?": site = str(e.input_shapes)[:90]
    k = (e.name, site); agg[k][0] += e.self_device_time_total; agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print("ATen self device time total %.2f ms" % (tot / 1e3))
for (n, s), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:130]:
    print("%8.3f ms %5d  %-28s %s" % (t / 1e3, c, n, s))
