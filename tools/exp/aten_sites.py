"""Which python lines of the package issue the ATen kernels of one training step: every ATen call that reaches the dispatcher is counted
with the innermost frame inside u2seg_amd (TorchDispatchMode; view / metadata ops launch nothing and are left out by name).
usage: python tools/exp/aten_sites.py"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from u2seg_amd.config import get_cfg  # noqa: E402
from u2seg_amd.data import make_synthetic_batch  # noqa: E402
from u2seg_amd.engine import SimpleTrainer  # noqa: E402
from u2seg_amd.modeling import build_model  # noqa: E402
from u2seg_amd.solver import build_lr_scheduler, build_optimizer  # noqa: E402

NO_KERNEL = ("view", "reshape", "expand", "permute", "transpose", "t.", "select", "slice", "unsqueeze", "squeeze", "detach", "alias",
             "as_strided", "unbind", "split", "chunk", "narrow", "size", "stride", "is_", "_unsafe_view", "empty", "numel", "dim",
             "sym_", "item", "_local_scalar_dense", "lift_fresh", "unfold", "set_", "resize_", "_to_copy_meta", "result_type",
             "can_cast", "record_stream", "_has_compatible", "_pin_memory", "is_pinned", "new_empty", "diagonal", "real", "view_as")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.calls = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(name.startswith(p) for p in NO_KERNEL):
            cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or "cuda" in str(kwargs)
            if cuda or name.startswith(("zeros", "ones", "full", "arange", "tensor", "scalar_tensor")):
                site = "?"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if "u2seg_amd" in fr.filename:
                        site = "%s:%d %s" % (fr.filename.split("u2seg_amd/")[-1], fr.lineno, fr.name)
                        break
                self.calls[(site, name)] += 1
        return func(*args, **(kwargs or {}))


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
spy = Spy()
with spy:
    trainer.run_step(batch)
torch.cuda.synchronize()
by_site = collections.Counter()
for (site, name), n in spy.calls.items():
    by_site[site] += n
print(sum(spy.calls.values()), "ATen calls with a device tensor in one step (upper bound of the launches; a few fuse or launch two)")
for site, n in by_site.most_common(60):
    ops = ", ".join("%s x%d" % (nm.split(".")[0], c) for (s, nm), c in sorted(spy.calls.items(), key=lambda kv: -kv[1]) if s == site)
    print("%4d  %-52s %s" % (n, site, ops[:150]))
