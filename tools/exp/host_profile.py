"""Host-side profile of the training step (cProfile over 8 steps after 4 warm-up steps): where the python time of the host-bound heads'
forward phase goes.  python tools/exp/host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
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
for _ in range(4):
    trainer.run_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(8):
    trainer.run_step(batch)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
st.print_stats(r"u2seg_amd", 70)
