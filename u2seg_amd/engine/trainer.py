"""Training step driver with the contract of detectron2's AMPTrainer.run_step (engine/train_loop.py:479-521):
forward -> dict of losses -> sum -> backward -> gradient exchange -> per-parameter clip + SGD -> LR schedule.

bf16 needs no GradScaler (the reference's fp16 path scales/unscales, train_loop.py:504-521).  One process per GPU;
gradients are summed with RCCL all-reduce over the flat gradient arena (solver/build.py) after backward."""
import argparse
import os

import torch
import torch.distributed as dist

from ..layers import functional as F


def launch_info():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


class SimpleTrainer:
    def __init__(self, model, optimizer, scheduler=None):
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.iter = 0
        self.last_losses = None
        # gradient exchange overlapped with backward: the model calls this when the gradients of the FPN outputs exist,
        # i.e. when every head parameter's gradient is final (the heads come after the backbone in the arena)
        heads = [m for name, m in model.named_children() if name != "backbone"]
        first = [p for m in heads for p in m.parameters() if p.requires_grad]
        if first and hasattr(optimizer, "begin_all_reduce_tail") and hasattr(model, "on_heads_backward_done"):
            tail = min(optimizer.offset_of(p) for p in first)
            back = [optimizer.offset_of(p) for p in model.backbone.parameters() if p.requires_grad]
            if not back or max(back) < tail:  # the heads really form the tail of the arena
                model.on_heads_backward_done = lambda: optimizer.begin_all_reduce_tail(tail)

    def run_step(self, batched_inputs):
        assert self.model.training, "[SimpleTrainer] model was changed to eval mode!"
        self.optimizer.zero_grad()
        loss_dict = self.model(batched_inputs)
        # train_loop.py:491 sums the dict; here one stack + one sum instead of a chain of len - 1 scalar adds (and no kernels in
        # backward: the gradient of every entry is a view of the same 1.0)
        vals = list(loss_dict.values())
        losses = torch.stack(vals).sum() if len(vals) > 1 and all(v.dim() == 0 and v.dtype == vals[0].dtype for v in vals) \
            else sum(vals)
        losses.backward()
        F.assert_no_deferred_gradients()
        grad_scale = self.optimizer.all_reduce_grads()
        self.optimizer.step(grad_scale)
        if self.scheduler is not None:
            self.scheduler.step()
        self.iter += 1
        self.last_losses = loss_dict
        return loss_dict

    def check_finite(self):
        """train_loop.py:411-415 raises on non-finite total loss; call off the critical path."""
        total = float(sum(v.detach() for v in self.last_losses.values()))
        if not (total == total and abs(total) != float("inf")):
            raise FloatingPointError("Loss became infinite or NaN at iteration={}!\nloss_dict = {}".format(
                self.iter, {k: float(v) for k, v in self.last_losses.items()}))
        return total


def default_argument_parser():
    """Flags of detectron2.engine.default_argument_parser (engine/defaults.py:82-144).  Deviation recorded in
    DESIGN.md: the reference edits --eval-only to default=True; here it defaults to False so training runs."""
    p = argparse.ArgumentParser(description="u2seg_amd training / evaluation")
    p.add_argument("--config-file", default="configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml", metavar="FILE")
    p.add_argument("--resume", action="store_true")
    p.add_argument("--eval-only", action="store_true")
    p.add_argument("--eval-mode", default="hungarian_matching")
    p.add_argument("--num-gpus", type=int, default=1)
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    p.add_argument("--dist-url", default="tcp://127.0.0.1:29500")
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER)
    return p
