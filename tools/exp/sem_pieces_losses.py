"""Loss dicts of the first training steps with the semantic head launched in pieces / as a whole (same seeds): python tools/exp/sem_pieces_losses.py"""
import os
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
batch = make_synthetic_batch(16, start_index=0, height=800, width=1333, device=dev)
for mode in ("1", "0", "1", "0"):
    os.environ["U2_SEM_PIECES"] = mode
    torch.manual_seed(1234)
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", dev, "SOLVER.IMS_PER_BATCH", 16])
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    trainer = SimpleTrainer(model, opt, build_lr_scheduler(cfg, opt))
    torch.manual_seed(1000)
    for it in range(4):
        ld = trainer.run_step(batch)
        torch.cuda.synchronize()
        print("pieces=%s step %d total %.6f  " % (mode, it, float(sum(ld.values()))) + " ".join("%s=%.5f" % (k[5:], float(v)) for k, v in ld.items()))
    del model, opt, trainer
