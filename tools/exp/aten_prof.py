"""Which Python lines launch the ATen / copy kernels of a training step (torch.profiler with stacks, one step)."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from u2seg_amd.config import get_cfg
from u2seg_amd.data import make_synthetic_batch
from u2seg_amd.engine import SimpleTrainer
from u2seg_amd.modeling import build_model
from u2seg_amd.solver import build_lr_scheduler, build_optimizer

dev = "cuda"
torch.manual_seed(1234)
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", dev, "SOLVER.IMS_PER_BATCH", 16])
model = build_model(cfg); model.train()
opt = build_optimizer(cfg, model)
trainer = SimpleTrainer(model, opt, build_lr_scheduler(cfg, opt))
batches = [make_synthetic_batch(16, start_index=i * 16, height=800, width=1333, device=dev) for i in range(2)]
for i in range(3):
    trainer.run_step(batches[i % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    trainer.run_step(batches[1])
    torch.cuda.synchronize()

# kernel events -> launching CPU op -> python stack
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    kt = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
    if not ev.kernels:
        continue
    names = [k.name for k in ev.kernels]
    if all(("u2" in n or "conv_" in n or "kernel(" in n) and "at::" not in n and "rocclr" not in n and "elementwise" not in n for n in names):
        continue
    frames = [f for f in (ev.stack or []) if "u2seg_amd" in f or "bench" in f or "engine" in f]
    site = " <- ".join(fr.split("/")[-1] for fr in frames[:3]) if frames else "(no stack)"
    key = (ev.name, site)
    agg[key][0] += len(ev.kernels)
    agg[key][1] += kt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot_t = sum(v[1] for _, v in rows); tot_n = sum(v[0] for _, v in rows)
print("non-u2 kernels of one step: %d launches, %.3f ms" % (tot_n, tot_t / 1e3))
for (name, site), (n, t) in rows[:70]:
    print("%7.1f us %4d  %-28s %s" % (t, n, name[:28], site[:200]))
