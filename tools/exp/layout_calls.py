"""Which weights still get their bf16 kernel layout from a per-launch u2_weight_layout call inside a training step (the rest is
rewritten by the optimizer's one batched launch).  usage: python tools/exp/layout_calls.py"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from u2seg_amd import _hip  # noqa: E402
from u2seg_amd.config import get_cfg  # noqa: E402
from u2seg_amd.data import make_synthetic_batch  # noqa: E402
from u2seg_amd.engine import SimpleTrainer  # noqa: E402
from u2seg_amd.modeling import build_model  # noqa: E402
from u2seg_amd.solver import build_lr_scheduler, build_optimizer  # noqa: E402

dev = "cuda:0"
torch.manual_seed(1234)
cfg = get_cfg()
cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", dev, "SOLVER.IMS_PER_BATCH", 16])
model = build_model(cfg)
model.train()
opt = build_optimizer(cfg, model)
trainer = SimpleTrainer(model, opt, build_lr_scheduler(cfg, opt))
batch = make_synthetic_batch(16, start_index=0, height=800, width=1333, device=dev)
for _ in range(2):
    trainer.run_step(batch)
torch.cuda.synchronize()
calls = collections.Counter()
orig = _hip.call


def spy(name, *a):
    if name == "u2_weight_layout":
        site = [f for f in traceback.extract_stack()[:-1] if "u2seg_amd" in f.filename and "functional.py" not in f.filename]
        where = "%s:%d" % (os.path.relpath(site[-1].filename, ROOT), site[-1].lineno) if site else "?"
        calls[(tuple(a[2:]), where)] += 1
    return orig(name, *a)


_hip.call = spy
import u2seg_amd.layers.functional as Fn  # noqa: E402
Fn._hip.call = spy
trainer.run_step(batch)
torch.cuda.synchronize()
print(sum(calls.values()), "u2_weight_layout launches in one step; (n, cin, taps, cp, npad, mode), call site, count:")
for (k, where), v in sorted(calls.items(), key=lambda kv: -kv[1]):
    print(k, where, v)
