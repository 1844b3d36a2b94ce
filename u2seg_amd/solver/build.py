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
        # per-parameter hyper-parameters by the reference's rule (solver/build.py:218-236 get_default_optimizer_params): the
        # module tree is walked in named_modules order, a normalisation layer's parameters take WEIGHT_DECAY_NORM, and the
        # legacy bias override - applied after it, to every parameter literally named "bias" - WEIGHT_DECAY_BIAS
        # `key_of` is the reference's grouping key: its per-parameter dict starts as {"lr": base_lr} - build_optimizer never
        # hands weight_decay to get_default_optimizer_params (solver/build.py:123-129), torch fills it in from the SGD default -
        # and gains a "weight_decay" entry only where an override applied.  So a parameter without override (key None) and one
        # whose override EQUALS the default still sit in different groups, and all overridden parameters of one value share one.
        wd_of, key_of, seen = {}, {}, set()
        for mod in model.modules():
            for pname, p in mod.named_parameters(recurse=False):
                if not p.requires_grad or id(p) in seen:
                    continue
                seen.add(id(p))
                key = None
                if isinstance(mod, (BatchNorm2d, GroupNorm)) and weight_decay_norm is not None:
                    key = float(weight_decay_norm)
                if pname == "bias" and weight_decay_bias is not None:
                    key = float(weight_decay_bias)
                key_of[id(p)] = key
                wd_of[id(p)] = float(weight_decay) if key is None else key
        self.params = [p for p in model.parameters() if p.requires_grad]
        assert len(self.params) == len(wd_of)
        # the reference merges parameters with equal hyper-parameter dicts into one torch param group, groups in order of first
        # appearance (reduce_param_groups, solver/build.py:255-279); torch numbers the parameters group after group.  That
        # numbering is the key of a reference checkpoint's optimizer state (state_dict / load_state_dict below).
        group_keys = list(dict.fromkeys(key_of[id(p)] for p in self.params))
        self.group_wd = [float(weight_decay) if k is None else k for k in group_keys]
        self.group_members = [[i for i, p in enumerate(self.params) if key_of[id(p)] == k] for k in group_keys]
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
            wds.append(wd_of[id(p)])
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
        self.base_lr = lr  # "initial_lr" of the param groups once a torch LR scheduler has touched them
        self.param_offset = []
        off = 0
        for p in self.params:
            self.param_offset.append(off)
            off += p.numel()
        self.bucket_elems = bucket_bytes // 4
        self._pending, self._tail_from = [], None
        self._exchange_stream, self._step_stream = None, None
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
        from ..layers.functional import reset_deferred_gradients

        reset_deferred_gradients()   # a backward pass that raised midway must not leave deferred gradients behind
        if self.flat_grad.is_cuda:
            self._step_stream = torch.cuda.current_stream(self.flat_grad.device)  # the stream the training step is issued on
        self.flat_grad.zero_()

    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _launch_all_reduce(self, lo, hi):
        """Async all-reduce of flat_grad[lo:hi] in buckets.  On the GPU the collectives are issued from a stream of their own
        that first waits - explicitly, event by event - for every stream that produced gradients: the stream the step runs on,
        the weight-gradient side stream and the branch streams.  ProcessGroupNCCL orders a collective behind whatever stream is
        current when it is called; the tail exchange is called from an autograd hook, where "current" is whichever stream the
        engine selected for that node, so nothing here depends on it."""
        if not self.flat_grad.is_cuda:
            for s in range(lo, hi, self.bucket_elems):
                self._pending.append(dist.all_reduce(self.flat_grad[s : min(s + self.bucket_elems, hi)], async_op=True))
            return
        from ..layers.functional import producer_streams

        dev = self.flat_grad.device
        if self._exchange_stream is None:
            self._exchange_stream = torch.cuda.Stream(device=dev)
        ex = self._exchange_stream
        producers = producer_streams(dev) + [torch.cuda.current_stream(dev)]
        if self._step_stream is not None:
            producers.append(self._step_stream)
        for st in producers:
            ev = torch.cuda.Event()
            ev.record(st)
            ex.wait_event(ev)
        with torch.cuda.stream(ex):
            for s in range(lo, hi, self.bucket_elems):
                self._pending.append(dist.all_reduce(self.flat_grad[s : min(s + self.bucket_elems, hi)], async_op=True))

    def begin_all_reduce_tail(self, first_elem):
        """Start summing the gradients of the arena's tail [first_elem, total) while backward is still running: the tail
        holds the head parameters (RPN, ROI heads, semantic head), which are final once the gradients of the FPN outputs
        exist; RCCL works on them over xGMI while the backbone backward (~40 % of the step) keeps the CUs busy."""
        if not self._distributed() or self._tail_from is not None:
            return
        self._tail_from = int(first_elem)  # (_launch_all_reduce waits for the producing streams itself)
        self._launch_all_reduce(self._tail_from, self.total)

    def all_reduce_grads(self):
        """Sum gradients over ranks in a few large buckets (mean is folded into the step's grad_scale)."""
        if not self._distributed():
            return 1.0
        self._launch_all_reduce(0, self.total if self._tail_from is None else self._tail_from)
        for h in self._pending:
            h.wait()  # the current stream (the optimizer's) waits for the collective; the host does not block on NCCL
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
        """The state dict `torch.optim.SGD` - the reference's optimizer (solver/build.py:119-139) - would write for this
        model: `param_groups` in the reference's grouping with torch's parameter numbering, `state[i]["momentum_buffer"]` a
        copy of parameter i's slice of the momentum arena.  A reference trainer can resume from it, and `load_state_dict`
        reads what a reference trainer wrote.  (Before the first step torch has no state entries; here the buffers exist
        from the start and are written as zeros, which is what torch's first step computes from: buf = grad.)"""
        groups, state, n = [], {}, 0
        for wd, members in zip(self.group_wd, self.group_members):
            groups.append({"lr": float(self.lr), "momentum": float(self.momentum), "dampening": 0, "weight_decay": wd,
                           "nesterov": False, "maximize": False, "foreach": True, "differentiable": False, "fused": None,
                           "initial_lr": float(self.base_lr), "params": list(range(n, n + len(members)))})
            for i in members:
                off = self.param_offset[i]
                p = self.params[i]
                state[n] = {"momentum_buffer": self.flat_mom[off : off + p.numel()].view(p.shape).clone()}
                n += 1
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "param_groups" not in sd:  # this package's first on-disk form: the whole momentum arena + lr
            self.flat_mom.copy_(sd["momentum"])
            self.lr = sd["lr"]
            return
        groups = sd["param_groups"]
        if [len(g["params"]) for g in groups] != [len(m) for m in self.group_members]:
            raise ValueError("loaded state dict has a different number of parameter groups / parameters per group: %s vs %s"
                             % ([len(g["params"]) for g in groups], [len(m) for m in self.group_members]))
        lrs = {float(g["lr"]) for g in groups}
        assert len(lrs) == 1, "per-group learning rates (BIAS_LR_FACTOR != 1) are not implemented"
        self.lr = lrs.pop()
        if all("initial_lr" in g for g in groups):  # written once a torch LR scheduler has touched the optimizer
            base = {float(g["initial_lr"]) for g in groups}
            assert len(base) == 1, base
            self.base_lr = base.pop()
        for gi, (g, members) in enumerate(zip(groups, self.group_members)):
            assert not g.get("nesterov", False) and not g.get("dampening", 0) and not g.get("maximize", False), g
            self.momentum = float(g.get("momentum", self.momentum))
            wd = float(g.get("weight_decay", self.group_wd[gi]))
            if wd != self.group_wd[gi]:
                # torch takes a group's hyper-parameters from the file, not from the constructor (Optimizer.load_state_dict);
                # groups are addressed by position: two groups may hold equal values
                self.group_wd[gi] = wd
                self.wd[torch.tensor(members, device=self.wd.device)] = wd
            for key, i in zip(g["params"], members):
                off, p = self.param_offset[i], self.params[i]
                buf = (sd["state"].get(key) or {}).get("momentum_buffer")
                dst = self.flat_mom[off : off + p.numel()]
                if buf is None:  # torch creates the buffer in the first step: no entry = no history
                    dst.zero_()
                    continue
                if tuple(buf.shape) != tuple(p.shape):
                    raise ValueError("momentum buffer %s of parameter %d does not fit %s" % (tuple(buf.shape), key, tuple(p.shape)))
                dst.copy_(buf.reshape(-1))


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
        # "last_epoch" / "base_lrs" are what the reference's scheduler objects read back (fvcore's LRMultiplier and the plain
        # WarmupMultiStepLR of solver/lr_scheduler.py are torch LRSchedulers: load_state_dict = __dict__.update)
        groups = len(getattr(self.optimizer, "group_members", [0]))
        return {"last_iter": self.last_iter, "last_epoch": self.last_iter, "base_lrs": [self.base_lr] * groups}

    def load_state_dict(self, state):
        self.resume_at(state["last_iter"] if "last_iter" in state else state["last_epoch"])


def build_lr_scheduler(cfg, optimizer):
    s = cfg.SOLVER
    assert s.LR_SCHEDULER_NAME == "WarmupMultiStepLR"
    steps = [x for x in s.STEPS if x <= s.MAX_ITER]
    return WarmupMultiStepLR(optimizer, s.BASE_LR, steps, s.GAMMA, s.WARMUP_FACTOR, min(s.WARMUP_ITERS, s.MAX_ITER),
                             s.WARMUP_METHOD)
