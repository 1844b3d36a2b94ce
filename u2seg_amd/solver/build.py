"""Optimizer and LR schedule (detectron2/solver/build.py:29-323, solver/lr_scheduler.py:22-138).

All trainable tensors live in one flat fp32 arena (parameters, gradients and momentum are views), so that
 * data-parallel gradient exchange is a handful of large RCCL all-reduces over xGMI instead of 248 small ones,
 * per-parameter L2 clipping (CLIP_TYPE "norm", applied per tensor as in solver/build.py:36-37,63-73) + SGD with
   momentum and weight decay is two kernel launches (u2_sgd_clip_step)."""
import bisect

import torch
import torch.distributed as dist

from .. import _hip
from ..layers.modules import BatchNorm2d, GroupNorm

CHUNK = 1 << 16


class FlatSGD:
    def __init__(self, model, lr, momentum=0.9, weight_decay=0.0, weight_decay_norm=0.0, weight_decay_bias=None,
                 clip_value=0.0, bucket_bytes=64 << 20):
        norm_params = set()
        for mod in model.modules():
            if isinstance(mod, (BatchNorm2d, GroupNorm)):
                norm_params.update(id(p) for p in mod.parameters(recurse=False))
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.total = total
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_mom = torch.zeros(total, dtype=torch.float32, device=dev)
        chunk_tensor, chunk_begin, chunk_len, wds, first_chunk = [], [], [], [], []
        off = 0
        for t, (name_p) in enumerate(self.params):
            p = name_p
            n = p.numel()
            self.flat_param[off : off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off : off + n].view(p.shape)
            p.grad = self.flat_grad[off : off + n].view(p.shape)
            p._u2_grad = p.grad  # layers/functional.py:grad_slot - kernels accumulate into the arena directly
            if id(p) in norm_params:
                wds.append(weight_decay_norm)
            elif p.dim() == 1 and weight_decay_bias is not None:
                wds.append(weight_decay_bias)
            else:
                wds.append(weight_decay)
            first_chunk.append(len(chunk_tensor))
            for s in range(0, n, CHUNK):
                chunk_tensor.append(t)
                chunk_begin.append(off + s)
                chunk_len.append(min(CHUNK, n - s))
            off += n
        self.chunk_tensor = torch.tensor(chunk_tensor, dtype=torch.int32, device=dev)
        self.chunk_begin = torch.tensor(chunk_begin, dtype=torch.int64, device=dev)
        self.chunk_len = torch.tensor(chunk_len, dtype=torch.int32, device=dev)
        self.wd = torch.tensor(wds, dtype=torch.float32, device=dev)
        first_chunk.append(len(chunk_tensor))
        self.first_chunk = torch.tensor(first_chunk, dtype=torch.int32, device=dev)
        self.partial = torch.zeros(len(chunk_tensor), dtype=torch.float32, device=dev)
        self.lr, self.momentum, self.clip_value = lr, momentum, clip_value
        self.bucket_elems = bucket_bytes // 4
        self._pending, self._tail_from = [], None
        # bf16 kernel layouts cached on the parameters (layers/functional.py:_weight_layout): [stamp] is shared with every
        # parameter; step() bumps it and rewrites all registered layouts with one launch.
        self._stamp = [0]
        self._layout_entries = []  # (param, key, entry)
        self._layout_table = None
        self._layout_table_len = -1
        for p in self.params:
            p._u2_stamp = self._stamp
            p._u2_layout_register = self._register_layout
            p.__dict__.pop("_u2_layouts", None)
            p.__dict__.pop("_u2_bias_rounded", None)

    def offset_of(self, param):
        """Element offset of a parameter inside the flat arena."""
        return (param.data_ptr() - self.flat_param.data_ptr()) // 4

    def zero_grad(self):
        self.flat_grad.zero_()

    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _launch_all_reduce(self, lo, hi):
        for s in range(lo, hi, self.bucket_elems):
            self._pending.append(dist.all_reduce(self.flat_grad[s : min(s + self.bucket_elems, hi)], async_op=True))

    def begin_all_reduce_tail(self, first_elem):
        """Start summing the gradients of the arena's tail [first_elem, total) while backward is still running: the tail
        holds the head parameters (RPN, ROI heads, semantic head), which are final once the gradients of the FPN outputs
        exist; RCCL works on them over xGMI while the backbone backward (~40 % of the step) keeps the CUs busy."""
        if not self._distributed() or self._tail_from is not None:
            return
        from ..layers.functional import join_all_streams

        join_all_streams()  # the heads' gradients were produced on the side stream and on their branches' streams
        self._tail_from = int(first_elem)
        self._launch_all_reduce(self._tail_from, self.total)

    def all_reduce_grads(self):
        """Sum gradients over ranks in a few large buckets (mean is folded into the step's grad_scale)."""
        if not self._distributed():
            return 1.0
        from ..layers.functional import join_all_streams

        join_all_streams()
        self._launch_all_reduce(0, self.total if self._tail_from is None else self._tail_from)
        for h in self._pending:
            h.wait()
        self._pending, self._tail_from = [], None
        return 1.0 / dist.get_world_size()

    def step(self, grad_scale=1.0):
        from ..layers.functional import join_all_streams

        join_all_streams()
        _hip.call("u2_sgd_clip_step", self.flat_param, self.flat_grad, self.flat_mom, self.chunk_tensor, self.chunk_begin,
                  self.chunk_len, self.chunk_tensor.numel(), self.partial, self.first_chunk, self.wd, float(self.lr),
                  float(self.momentum), float(self.clip_value), float(grad_scale))
        self.refresh_layouts()

    def _register_layout(self, param, key, entry):
        self._layout_entries.append((param, key, entry))

    def refresh_layouts(self):
        """The parameters changed under the kernels' feet: new stamp, then rewrite every cached layout in one launch."""
        self._stamp[0] += 1
        ents = self._layout_entries
        if not ents:
            return
        if self._layout_table_len != len(ents):
            import numpy as np

            desc = np.zeros(len(ents), dtype=np.dtype([("src", "<i8"), ("dst", "<u8"), ("N", "<i4"), ("Cin", "<i4"), ("T", "<i4"),
                                                       ("Cp", "<i4"), ("Npad", "<i4"), ("mode", "<i4"), ("blk", "<i4"),
                                                       ("rsv", "<i4")], align=True))
            assert desc.dtype.itemsize == 48
            base = self.flat_param.data_ptr()
            blocks = 0
            for i, (p, key, ent) in enumerate(ents):
                off = p.data_ptr() - base
                assert 0 <= off < self.total * 4 and off % 4 == 0
                desc[i] = (off // 4, ent[0].data_ptr()) + tuple(key) + (blocks, 0)
                n_, cin_, t_, cp_, npad_, mode_ = key
                if mode_ == 3:
                    blocks += (n_ + 4095) // 4096             # a bias vector rounded through bf16, fp32 out
                elif mode_ == 0 and t_ == 1 and cp_ == cin_:
                    blocks += (n_ * cp_ + 4095) // 4096  # plain conversion (u2_weight_layout_batched: the same rule)
                elif mode_ == 0:
                    blocks += n_ * ((cp_ + 63) // 64)
                else:
                    blocks += ((npad_ + 63) // 64) * ((cin_ * t_ + 63) // 64)
            self._layout_table = torch.from_numpy(desc.view(np.uint8).copy()).to(self.flat_param.device)
            self._layout_table_len = len(ents)
            self._layout_blocks = blocks
        _hip.call("u2_weight_layout_batched", self.flat_param, self._layout_table, len(ents), self._layout_blocks)
        stamp = self._stamp[0]
        for p, _, ent in ents:
            ent[1], ent[2] = p._version, stamp

    def state_dict(self):
        return {"momentum": self.flat_mom.clone(), "lr": self.lr}

    def load_state_dict(self, sd):
        self.flat_mom.copy_(sd["momentum"])
        self.lr = sd["lr"]


def build_optimizer(cfg, model):
    s = cfg.SOLVER
    clip = 0.0
    if s.CLIP_GRADIENTS.ENABLED:
        assert s.CLIP_GRADIENTS.CLIP_TYPE == "norm" and float(s.CLIP_GRADIENTS.NORM_TYPE) == 2.0, \
            "only per-parameter L2 norm clipping (the U2Seg setting) is implemented"
        clip = float(s.CLIP_GRADIENTS.CLIP_VALUE)
    assert float(s.BIAS_LR_FACTOR) == 1.0 and not s.NESTEROV
    return FlatSGD(model, lr=s.BASE_LR, momentum=s.MOMENTUM, weight_decay=s.WEIGHT_DECAY,
                   weight_decay_norm=s.WEIGHT_DECAY_NORM, weight_decay_bias=s.WEIGHT_DECAY_BIAS, clip_value=clip)


class WarmupMultiStepLR:
    """lr(iter) = base_lr * gamma^(#milestones passed) * warmup(iter) with linear warm-up from warmup_factor
    (solver/build.py:283-323 via fvcore's MultiStepParamScheduler + LinearParamScheduler composite)."""

    def __init__(self, optimizer, base_lr, milestones, gamma, warmup_factor, warmup_iters, warmup_method="linear"):
        assert warmup_method == "linear"
        self.optimizer, self.base_lr = optimizer, base_lr
        self.milestones, self.gamma = sorted(milestones), gamma
        self.warmup_factor, self.warmup_iters = warmup_factor, warmup_iters
        self.last_iter = 0
        self.optimizer.lr = self.get_lr(0)

    def get_lr(self, it):
        value = self.base_lr * self.gamma ** bisect.bisect_right(self.milestones, it)
        if it < self.warmup_iters:
            # fvcore composite: linear from warmup_factor*start to the multistep value at the end of warm-up
            end = self.base_lr * self.gamma ** bisect.bisect_right(self.milestones, self.warmup_iters)
            start = self.warmup_factor * self.base_lr
            alpha = it / self.warmup_iters
            value = start * (1 - alpha) + end * alpha
        return value

    def step(self):
        self.last_iter += 1
        self.optimizer.lr = self.get_lr(self.last_iter)

    def resume_at(self, iteration):
        """Continue the schedule at `iteration` (the next step to run): the reference restores its scheduler through the
        checkpointer (engine/defaults.py:410-421 resume_or_load -> start_iter); without this a resumed run would restart the
        warm-up and count the STEPS milestones from the resume point."""
        self.last_iter = int(iteration)
        self.optimizer.lr = self.get_lr(self.last_iter)

    def state_dict(self):
        return {"last_iter": self.last_iter}

    def load_state_dict(self, state):
        self.resume_at(state["last_iter"])


def build_lr_scheduler(cfg, optimizer):
    s = cfg.SOLVER
    assert s.LR_SCHEDULER_NAME == "WarmupMultiStepLR"
    steps = [x for x in s.STEPS if x <= s.MAX_ITER]
    return WarmupMultiStepLR(optimizer, s.BASE_LR, steps, s.GAMMA, s.WARMUP_FACTOR, min(s.WARMUP_ITERS, s.MAX_ITER),
                             s.WARMUP_METHOD)
