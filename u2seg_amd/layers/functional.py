"""torch.autograd.Function wrappers over the C-ABI HIP launchers (include/u2seg_hip.h).

Activations are NHWC bfloat16 tensors ``[B, H, W, C]`` whose physical channel count is a multiple
of 32 (logical channel counts such as 28 / 3 / 12 / 801 / 4 are zero padded on the right and the
consumer is told the logical width).  Parameters stay fp32 in the reference layout
(``[Cout, Cin, kh, kw]``, state-dict compatible with detectron2) and are re-laid-out to the kernel
layout (``[Cout][kh*kw][Cin]`` bf16) on the fly.

Every function here launches HIP kernels; none has a CPU or ATen compute fallback.
"""
import ctypes
import math

import functools
import os

import torch
import torch.distributed as dist
from torch.autograd import Function

from .. import _hip

BF16 = torch.bfloat16


def ceil32(n):
    return (n + 31) // 32 * 32


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _ZeroPool:
    """Small zero-initialised fp32 scratch tensors (statistics, split-K gradient accumulators) carved out of 16 MB
    chunks: one fill kernel per chunk instead of one per tensor.  Slices are handed out once and never recycled."""

    CHUNK = 1 << 22

    def __init__(self):
        self.chunks = {}  # (device, stream) -> [chunk, offset]: a chunk is filled on, and only handed out to, one stream

    def take(self, shape, device):
        numel = 1
        for d in shape:
            numel *= d
        if numel * 2 > self.CHUNK or numel == 0:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        n = (numel + 63) // 64 * 64
        key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        ent = self.chunks.get(key)
        if ent is None or ent[1] + n > self.CHUNK:
            ent = self.chunks[key] = [torch.zeros(self.CHUNK, dtype=torch.float32, device=device), 0]
        v = ent[0][ent[1] : ent[1] + numel].view(shape)
        ent[1] += n
        return v


_zero_pool = _ZeroPool()


def zeros_f32(shape, device):
    return _zero_pool.take(tuple(shape), torch.device(device))


def release_library_scratch():
    """Frees the device memory libu2seg_hip.so owns (stream-K hand-over slots, weight-gradient partial tiles: include/u2seg_hip.h
    u2_release_scratch); it is allocated again by the next launch that needs it.  Call it next to torch.cuda.empty_cache()."""
    return _hip.call_nostream("u2_release_scratch")


def _check_act(x):
    assert x.is_cuda and x.dtype == BF16 and x.is_contiguous() and x.dim() == 4 and x.shape[3] % 32 == 0, (
        "expected contiguous NHWC bf16 activation with C %% 32 == 0, got %s %s" % (tuple(x.shape), x.dtype)
    )


# --------------------------------------------------------------------------------------------
# weight layouts
# --------------------------------------------------------------------------------------------
def _weight_layout(w, cp, npad, mode, owner=None):
    """bf16 kernel layout of an fp32 [N, Cin, KH, KW] weight.  `owner` is the nn.Parameter whose memory `w` is (w itself or a
    reshaped view of it).  Parameters owned by solver.FlatSGD keep their layouts: the optimizer rewrites all of them in
    one launch right after each step (u2_weight_layout_batched), so the training passes launch no layout kernels.  A cached
    layout is trusted only while the parameter's autograd version and the optimizer's step stamp are unchanged."""
    n, cin, kh, kw = w.shape
    t = kh * kw
    shape = (n, t, cp) if mode == 0 else ((cp, t, npad) if mode == 1 else (t * cp, 1, npad))
    tcache = getattr(owner, "_u2_step_layouts", None) if owner is not None else None
    if tcache is not None:
        # a derived weight that lives for one pass (the RPN's concatenated predictor, used on five levels): layouts kept on the
        # tensor itself; the caller never modifies it in place
        key = (n, cin, t, cp, npad, mode)
        out = tcache.get(key)
        if out is None:
            out = tcache[key] = torch.empty(shape, dtype=BF16, device=w.device)
            _hip.call("u2_weight_layout", w.detach().float().contiguous(), out, n, cin, t, cp, npad, mode)
        return out
    stamp_ref = getattr(owner, "_u2_stamp", None) if owner is not None else None
    if mode != 0 and cp != cin:
        stamp_ref = None  # the batched transposer has no all-zero output rows; such layouts are built on the fly
    ent = None
    if stamp_ref is not None:
        cache = owner.__dict__.setdefault("_u2_layouts", {})
        key = (n, cin, t, cp, npad, mode)
        ent = cache.get(key)
        if ent is not None and ent[1] == owner._version and ent[2] == stamp_ref[0]:
            return ent[0]
    elif owner is not None and not torch.is_grad_enabled():
        # inference without an optimizer: the weights only change through torch (load_state_dict, copy_), which bumps the
        # version counter - one layout launch per weight for the whole evaluation instead of one per forward pass
        ecache = owner.__dict__.setdefault("_u2_eval_layouts", {})
        key = (n, cin, t, cp, npad, mode)
        hit = ecache.get(key)
        if hit is not None and hit[1] == owner._version and hit[2] == owner.data_ptr():
            return hit[0]
        out = torch.empty(shape, dtype=BF16, device=w.device)
        _hip.call("u2_weight_layout", w.detach().float().contiguous(), out, n, cin, t, cp, npad, mode)
        ecache[key] = (out, owner._version, owner.data_ptr())
        return out
    out = ent[0] if ent is not None else torch.empty(shape, dtype=BF16, device=w.device)
    _hip.call("u2_weight_layout", w.detach().float().contiguous(), out, n, cin, t, cp, npad, mode)
    if stamp_ref is not None:
        if ent is None:
            ent = cache[key] = [out, owner._version, stamp_ref[0]]
            owner._u2_layout_register(owner, key, ent)
        else:
            ent[1], ent[2] = owner._version, stamp_ref[0]
    return out


def weight_fwd_layout(w, cp, owner=None):
    """[N, Cin, KH, KW] fp32 -> [N, KH*KW, cp] bf16 (channels zero padded to cp)."""
    return _weight_layout(w, cp, 0, 0, owner)


def weight_dgrad_layout(w, cp, npad, owner=None):
    """[N, Cin, KH, KW] fp32 -> [cp, KH*KW, npad] bf16 with the filter flipped in both spatial axes."""
    return _weight_layout(w, cp, npad, 1, owner)


def weight_fc_dgrad_layout(w, cp, npad, owner=None):
    """[N, Cin, KH, KW] fp32 -> [KH*KW*cp, 1, npad] bf16: dx[(kh,kw,c)] = sum_n dz[n] W[n, c, kh, kw]."""
    return _weight_layout(w, cp, npad, 2, owner)


# --------------------------------------------------------------------------------------------
# weight gradients on a second stream
# --------------------------------------------------------------------------------------------
# Nothing in the backward pass reads a weight gradient, so the wgrad kernels of all convolutions run on a side stream:
# they overlap with the data-gradient convs and - more usefully, MFMA work beside HBM-bound work - with the norm /
# ROI backward passes of the main stream, and one fills the other's partially occupied tail rounds.  The side stream
# waits for the main stream at every launch (its operands are produced there); the main stream waits for the side stream
# once, in a callback that autograd runs when the backward pass ends (so `.grad` is complete when backward() returns),
# and wherever gradients are consumed earlier (solver.FlatSGD: the all-reduce of the arena's tail starts inside backward).
_WGRAD_SIDE = os.environ.get("U2_WGRAD_SIDE_STREAM", "1") != "0"
_AUX_ENABLED = True


def set_stream_overlap(on):
    """Switch the side stream (weight gradients) and the second compute stream (semantic head) on / off at run time; off =
    every kernel of the step in one stream, one after the other (bench.py times its kernels that way)."""
    global _WGRAD_SIDE, _AUX_ENABLED
    _WGRAD_SIDE = bool(on) and os.environ.get("U2_WGRAD_SIDE_STREAM", "1") != "0"
    _AUX_ENABLED = bool(on)
_side_streams = {}
_side_dirty = {}


def _side_stream(device):
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device, priority=int(os.environ.get("U2_SIDE_PRIORITY", "0")))
    return s


_aux_streams = {}


def _cuda_device(device):
    device = torch.device(device)
    return torch.device("cuda", torch.cuda.current_device()) if device.index is None else device


def aux_stream(device, index=0):
    """A further compute stream for an independent branch of the model (None when switched off: U2_AUX_STREAM=0).
    index 0: the semantic head; index 1: the mask head beside the box cascade."""
    if not _AUX_ENABLED or os.environ.get("U2_AUX_STREAM", "1") == "0":
        return None
    key = (_cuda_device(device), index)
    s = _aux_streams.get(key)
    if s is None:
        s = _aux_streams[key] = torch.cuda.Stream(device=key[0], priority=int(os.environ.get("U2_AUX_PRIORITY", "0")))
    return s


def join_aux_stream(device, index=0):
    """The current stream waits for everything queued on that further compute stream."""
    if not torch.cuda.is_available() or torch.device(device).type != "cuda":
        return
    s = _aux_streams.get((_cuda_device(device), index))
    if s is not None:
        torch.cuda.current_stream(s.device).wait_stream(s)


# ---- where the stream-K form of the tile kernel may be used (round 6) ----
# Found with tools/exp/hang_hunt.sh: the driver's bench command hung in 4 of 23 runs (the chip never finished; the host sat in the
# step's first synchronisation), in 0 of 24 with the stream-K form forbidden, in 0 of 40 with the branch streams (semantic head, mask
# head) switched off, and still in 3 of 30 with the stream-K launches of different streams ordered behind each other by events.  So
# ONE stream-K launch - whose owners spin for peers that the dispatcher has yet to place on a CU, conv_tile.hip - can stall for good
# while kernels of a branch stream compete for the CUs; the mechanism is not understood (the weight-gradient side stream beside
# stream-K launches: 0 of 20).  Until it is, the product path asks for the stream-K form only where no branch stream has work in
# flight: inside the bottom-up backbone (forward: nothing else has been launched yet; backward: the heads' streams have been
# joined by the FPN's gradient fan-in), and anywhere when the branch streams are off (set_stream_overlap(False)).  Elsewhere the
# calls carry variant bit 28 (whole tiles).  U2_STREAMK_EVERYWHERE=1 restores the old behaviour.
_NO_STREAMK = 1 << 28
_SK_REGION = [0]


class streamk_region:
    """with streamk_region(): the convolutions created inside (and their backward passes) may take the stream-K form."""

    def __enter__(self):
        _SK_REGION[0] += 1

    def __exit__(self, *exc):
        _SK_REGION[0] -= 1
        return False


def _several_ranks():
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _conv_variant():
    if os.environ.get("U2_STREAMK_EVERYWHERE", "0") == "1" or "U2_CONV_VARIANT" in os.environ:
        return 0   # (U2_CONV_VARIANT: tests and A/B runs steer the library's dispatch themselves; it only applies to variant 0)
    if _several_ranks():
        # the gradient exchange of the arena's tail (RCCL kernels on their own stream, solver/build.py:begin_all_reduce_tail) runs
        # beside the backbone's backward pass: another stream competing for the CUs - never measured, so not risked
        return _NO_STREAMK
    if _SK_REGION[0] > 0 or not _AUX_ENABLED or os.environ.get("U2_AUX_STREAM", "1") == "0":
        return 0
    return _NO_STREAMK


# ---- an independent branch issued in pieces where the main chain is host-bound (round 6) ----
# The proposal bookkeeping of the ROI heads (match / relabel / sample between the cascade stages: ~40 tiny launches and one host
# synchronisation each) leaves the chip idle for 0.6-1.3 ms at a time while the host catches up (tools/step_idle.py: 3.7 ms of a 58 ms
# step).  A branch that only shares the FPN maps with it - the semantic head - is therefore not launched as a whole beside the RPN
# but handed over as a list of pieces; the bookkeeping code calls issue_deferred_piece() right before it becomes host-bound, and the
# piece's kernels (on the branch's own stream) run while the host synchronises and launches the small stuff.
_deferred_pieces = []


def defer_pieces(pieces):
    """Queue closures (each launches one piece of an independent branch on that branch's stream)."""
    _deferred_pieces.extend(pieces)


def issue_deferred_piece(n=1):
    """Launch the next n queued pieces (no-op when nothing is queued)."""
    while n > 0 and _deferred_pieces:
        _deferred_pieces.pop(0)()
        n -= 1


def flush_deferred():
    """Launch whatever is still queued (the owner of the branch calls this before it joins the branch's stream)."""
    issue_deferred_piece(len(_deferred_pieces))


def clear_deferred():
    """Drop queued pieces (a forward pass that raised midway must not leave its pieces to the next one)."""
    del _deferred_pieces[:]


def join_all_streams():
    """The current stream waits for the side stream and every further compute stream: used where gradients are consumed
    (optimizer step, gradient all-reduce - also the one of the arena's tail that starts inside the backward pass, when
    the heads' parameter gradients may still be in flight on the streams their branches ran on)."""
    join_wgrad_stream()
    for (dev, _index), s in _aux_streams.items():
        torch.cuda.current_stream(dev).wait_stream(s)


def producer_streams(device):
    """Every stream of this module that may hold gradient-producing work for `device` (weight-gradient side stream, the
    further compute streams); the caller adds the stream it runs the step on."""
    device = _cuda_device(device)
    out = [s for dev, s in _side_streams.items() if _cuda_device(dev) == device]
    out += [s for (dev, _index), s in _aux_streams.items() if dev == device]
    return out


def join_wgrad_stream(device=None):
    """Make the current stream wait for every weight-gradient launch issued so far."""
    for dev, s in _side_streams.items():
        if device is None or dev == torch.device(device):
            _side_dirty[dev] = False
            torch.cuda.current_stream(dev).wait_stream(s)


def _run_wgrad(device, operands, launch):
    """launch() (kernel launches + torch ops) on the side stream; `operands` were produced on the current stream."""
    if not _WGRAD_SIDE:
        launch()
        return
    main = torch.cuda.current_stream(device)
    side = _side_stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        launch()
    for t in operands:
        t.record_stream(side)  # their memory may be released on the main stream while the side stream still reads it
    _side_dirty[device] = True

    def _join():  # runs when the backward pass has finished (queued once per launch: the first one to run does the work)
        if _side_dirty.get(device):
            _side_dirty[device] = False
            torch.cuda.current_stream(device).wait_stream(side)  # the stream backward() was called on

    torch.autograd.Variable._execution_engine.queue_callback(_join)


# --------------------------------------------------------------------------------------------
# convolution / linear
# --------------------------------------------------------------------------------------------
_DET_STATS = False


def set_deterministic_stats(on):
    """Test switch: the BN column statistics that the conv epilogue accumulates with fp32 atomics (their order varies from
    run to run, so the forward pass differs in the last bits between runs) are taken from the stored output by a fixed-order
    reduction instead.  Free-running comparisons (tests/test_gpu_parity.py) use it to be reproducible."""
    global _DET_STATS
    _DET_STATS = bool(on)


def _fixed_order_stats(out, n):
    o = out[..., :n].float().reshape(-1, n)
    packed = torch.cat([o.sum(0), (o * o).sum(0), o.new_zeros(1)])
    st = packed[: 2 * n].view(2, n)
    st._u2_packed = packed
    return st


def _stats_buffer(n, device):
    """[sum | sum of squares] of a conv output's channels as a view of a zeroed [2n + 1] buffer: under SyncBN the extra slot
    takes this rank's element count and the whole buffer is all-reduced as it stands (no concatenation, no host round trip).
    The buffer is CONSUMED by batch_norm_act under SyncBN: after it the `stats` a caller still holds are the sums over all ranks,
    not this rank's (the only consumer in this package is the norm that follows the conv; clone before the norm if both are needed)."""
    packed = zeros_f32((2 * n + 1,), device)
    st = packed[: 2 * n].view(2, n)
    st._u2_packed = packed
    return st


class _Conv2dFn(Function):
    """F.conv2d (+bias)(+ReLU) with optional per-channel sum / sum-of-squares of the output
    (reference: detectron2/layers/wrappers.py:127-134)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu, want_stats, param_ref=None, round_bias=True):
        _check_act(x)
        # boxed so that autograd does not treat them as inputs: the owning Parameter and its arena gradient slice (looked up by
        # conv2d() - Function.forward runs with autograd disabled, where grad_slot() answers None)
        param = param_ref[0] if param_ref is not None else None
        wgrad_dst = param_ref[1] if param_ref is not None and len(param_ref) > 1 else None
        n, cin, kh, kw = weight.shape
        b, h, w_, cp = x.shape
        assert cin <= cp
        ho = (h + 2 * pad - kh) // stride + 1
        wo = (w_ + 2 * pad - kw) // stride + 1
        npad = ceil32(n)
        wk = weight_fwd_layout(weight, cp, param)
        alloc = torch.zeros if npad != n else torch.empty
        out = alloc((b, ho, wo, npad), dtype=BF16, device=x.device)
        stats = _stats_buffer(n, x.device) if want_stats else None
        # autocast casts every floating-point argument of a convolution, the bias included (pinned by the reference run under
        # bf16 autocast, tests/golden/bf16_units_golden.npz); a folded eval-mode norm shift is not a conv bias and stays fp32
        bias_f = None
        if bias is not None:
            bias_f = _conv_bias(bias, round_bias)
        ctx.variant = _conv_variant()
        _hip.call("u2_conv_igemm", x, wk, out, bias_f, None if _DET_STATS else stats, b, h, w_, cp, cp, ho, wo, n, npad, kh, kw,
                  pad, pad, stride, 1, int(relu), 0, ctx.variant)
        if want_stats and _DET_STATS:
            stats = _fixed_order_stats(out, n)
        ctx.save_for_backward(x, weight, out if relu else None)
        ctx.cfg = (stride, pad, relu, bias is not None)
        # 1x1 / linear weights: the gradient is accumulated straight into the optimizer's arena slice
        ctx.wgrad_dst = wgrad_dst if (kh * kw == 1 and wgrad_dst is not None) else None
        # the arena slice in the parameter's own [N, Cin, KH, KW] shape (multi-tap filters accumulate into it with a torch add)
        # (a reshaped view of the parameter - the box head's fc1 as a 7x7 conv - gets the slice in the view's shape)
        ctx.arena = None
        if wgrad_dst is not None and wgrad_dst.numel() == weight.numel() and wgrad_dst.is_contiguous():
            ctx.arena = wgrad_dst if tuple(wgrad_dst.shape) == tuple(weight.shape) else wgrad_dst.view(weight.shape)
        ctx.param = param
        # the bias gradient goes straight into the bias' arena slice as well (u2_colsum_add)
        ctx.bias_dst = param_ref[2] if param_ref is not None and len(param_ref) > 2 and bias is not None else None
        ctx.set_materialize_grads(False)  # no zero tensor for the (non-differentiable) statistics output
        if want_stats:
            ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dout, _dstats):
        if dout is None:
            return (None,) * 9
        x, weight, out = ctx.saved_tensors
        stride, pad, relu, has_bias = ctx.cfg
        n, cin, kh, kw = weight.shape
        b, h, w_, cp = x.shape
        _, ho, wo, npad = dout.shape
        lazy = _LAZY_GRADS.pop((dout.device.index, dout.data_ptr()), None) if _LAZY_GRADS else None
        if lazy is not None:
            # the gradient is still (dz, y, coefficients) of the batch normalisation behind this layer
            _ph, lz_dz, lz_y, lz_k = lazy
            assert not relu and not has_bias and kh == 1 and kw == 1 and stride == 1 and pad == 0
            dst = ctx.wgrad_dst
            wd = weight_dgrad_layout(weight, cp, npad, ctx.param)
            dx = torch.empty_like(x)
            if dst is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and _hip.call_status(
                    "u2_conv1x1_bwd_fused_bn", x, lz_dz, lz_y, lz_k[2], lz_k[3], lz_k[4], wd, dx, dst, b * h * w_, cp, cp, npad, npad,
                    npad, cp, n, cin, cin, 1, 0) == 0:
                return dx, None, None, None, None, None, None, None, None
            dout = torch.empty_like(lz_y)   # not served after all: the apply step as its own launch, then the usual paths
            _hip.call("u2_norm_bwd_apply", lz_dz, None, lz_y, lz_k[2], lz_k[3], lz_k[4], dout, None, 1, b * h * w_, npad, npad, 0,
                      None, None)
            dx = None
        dout = dout.contiguous()
        bias_done = False
        if relu:
            dz = torch.empty_like(dout)
            bdst = ctx.bias_dst if (has_bias and ctx.needs_input_grad[2]) else None
            if bdst is not None and bdst.is_contiguous() and bdst.dtype == torch.float32 and bdst.numel() == n:
                # ReLU backward and the bias gradient (into its arena slice) in one pass over dout / out
                _hip.call("u2_relu_bwd_colsum", dout, out, dz, bdst, zeros_f32((npad,), dout.device), b * ho * wo, npad, npad, n)
                bias_done = True
            else:
                _hip.call("u2_relu_bwd", dout, out, dz, dout.numel())
        else:
            dz = dout
        dx = dw = db = None
        fused = False
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and ctx.wgrad_dst is not None and kh == 1 and kw == 1 \
                and stride == 1 and pad == 0 and n > 128 and b * h * w_ >= FUSED_BWD_MIN_PIXELS:
            # both gradients of an expanding 1x1 layer in one pass over dz (u2_conv1x1_bwd_fused: dz is the large operand); the
            # launcher answers 1 for a shape it does not serve
            dst = ctx.wgrad_dst
            assert dst.is_contiguous() and dst.dtype == torch.float32 and dst.numel() == n * cin
            wd = weight_dgrad_layout(weight, cp, npad, ctx.param)
            dx = torch.empty_like(x)
            fused = _hip.call_status("u2_conv1x1_bwd_fused", x, dz, wd, dx, dst, b * h * w_, cp, cp, npad, npad, npad, cp, n, cin,
                                     cin, 1, 0) == 0
        if fused:
            pass
        elif ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if ho == 1 and wo == 1 and pad == 0 and h == kh and w_ == kw and kh * kw > 1:
                # "fully connected" conv (box head fc1): every input pixel meets exactly one tap, so the data gradient
                # is the plain GEMM dx[b, (kh,kw,c)] = dz[b, :] . W[:, (kh,kw,c)]
                wt = weight_fc_dgrad_layout(weight, cp, npad, ctx.param)
                _hip.call("u2_conv_igemm", dz, wt, dx, None, None, 1, b, 1, npad, npad, b, 1, kh * kw * cp, kh * kw * cp,
                          1, 1, 0, 0, 1, 1, 0, 0, ctx.variant)
            else:
                wd = weight_dgrad_layout(weight, cp, npad, ctx.param)
                _hip.call("u2_conv_igemm", dz, wd, dx, None, None, b, ho, wo, npad, npad, h, w_, cp, cp, kh, kw,
                          kh - 1 - pad, kw - 1 - pad, 1, stride, 0, 0, ctx.variant)
        if ctx.needs_input_grad[1] and not fused:
            dst = ctx.wgrad_dst
            arena = ctx.arena
            if dst is not None:
                assert dst.is_contiguous() and dst.dtype == torch.float32 and dst.numel() == n * cin

                def launch():
                    _hip.call("u2_conv_wgrad_into", x, dz, dst, b, h, w_, cp, cp, ho, wo, npad, npad, kh, kw, pad, pad, stride,
                              n, cin, cin, 1, 1, 0)

                _run_wgrad(x.device, (x, dz), launch)
            elif arena is not None and _WGRAD_SIDE:
                # multi-tap filters: kernel-layout scratch, then permuted into the arena slice - all of it on the side stream
                def launch():
                    # scratch from the (per-stream) zero pool, accumulated into the arena by one transposing kernel: the
                    # strided torch add took 70-170 us per 3x3 layer, 1.3 ms per step
                    dwk = zeros_f32((npad, kh * kw, cp), x.device)
                    _hip.call("u2_conv_wgrad", x, dz, dwk, b, h, w_, cp, cp, ho, wo, npad, npad, kh, kw, pad, pad, stride, 0)
                    if kh * kw * (cp + 1) * 4 <= 64 * 1024 and arena.is_contiguous():
                        _hip.call("u2_wgrad_permute_add", dwk, arena, n, cin, kh * kw, cp)
                    else:
                        arena.add_(dwk[:n, :, :cin].view(n, kh, kw, cin).permute(0, 3, 1, 2))

                _run_wgrad(x.device, (x, dz), launch)
            else:
                dwk = zeros_f32((npad, kh * kw, cp), x.device)
                _hip.call("u2_conv_wgrad", x, dz, dwk, b, h, w_, cp, cp, ho, wo, npad, npad, kh, kw, pad, pad, stride, 0)
                dw = dwk[:n, :, :cin].view(n, kh, kw, cin).permute(0, 3, 1, 2)
        if has_bias and ctx.needs_input_grad[2] and not bias_done:
            bdst = ctx.bias_dst
            if bdst is not None and bdst.is_contiguous() and bdst.dtype == torch.float32 and bdst.numel() == n:
                _hip.call("u2_colsum_add", dz, bdst, b * ho * wo, npad, npad, n)
            else:
                sums = zeros_f32((1, 2, npad), x.device)
                _hip.call("u2_colstats", dz, sums, 1, b * ho * wo, npad, npad)
                db = sums[0, 0, :n]
        return dx, dw, db, None, None, None, None, None, None


def _conv_bias(bias, round_bias):
    """The fp32 bias vector the conv epilogue adds (rounded through bf16 as autocast rounds it).  For a parameter owned by
    solver.FlatSGD the rounded copy is kept and rewritten by the optimizer's batched layout launch after every step (two small
    cast launches per biased conv and pass otherwise: 64 per training step)."""
    step_cache = getattr(bias, "_u2_step_bias", None)
    if step_cache is not None and round_bias:
        # a derived bias that lives for one pass (the RPN's concatenated predictor bias, used on five levels): rounded once
        v = step_cache.get("v")
        if v is None:
            v = step_cache["v"] = bias.detach().bfloat16().float().contiguous()
        return v
    stamp = getattr(bias, "_u2_stamp", None)
    if stamp is None or not round_bias:
        return (bias.detach().bfloat16().float() if round_bias else bias.detach().float()).contiguous()
    ent = bias.__dict__.get("_u2_bias_rounded")
    if ent is None:
        # registered with the optimizer's layout table (mode 3): rewritten with the weight layouts in the one launch after each
        # step (it was two cast launches per biased conv and step: 40 small launches)
        ent = [torch.empty(bias.shape, dtype=torch.float32, device=bias.device), -1, -1]
        bias.__dict__["_u2_bias_rounded"] = ent
        reg = getattr(bias, "_u2_layout_register", None)
        if reg is not None and bias.dim() == 1:
            reg(bias, (bias.numel(), 1, 1, 1, 0, 3), ent)
    if ent[1] != bias._version or ent[2] != stamp[0]:
        ent[0].copy_(bias.detach().bfloat16().float())
        ent[1], ent[2] = bias._version, stamp[0]
    return ent[0]


def grad_slot(param):
    """The optimizer's flat-arena gradient view of a parameter (solver/build.py:FlatSGD), or None.  Kernels that can
    accumulate into it directly do so and report no gradient to autograd (no temporary, no AccumulateGrad add)."""
    return getattr(param, "_u2_grad", None) if torch.is_grad_enabled() and param.requires_grad else None


def conv2d(x, weight, bias=None, stride=1, pad=0, relu=False, want_stats=False, param=None, round_bias=True):
    """`param`: the nn.Parameter that owns `weight`'s memory when `weight` is a reshaped view of it (defaults to `weight`
    itself when that is a Parameter); it carries the optimizer's gradient slot and the cached kernel layouts."""
    if param is None and isinstance(weight, torch.nn.Parameter):
        param = weight
    slot = grad_slot(param) if param is not None else None
    bslot = grad_slot(bias) if isinstance(bias, torch.nn.Parameter) else None
    out, stats = _Conv2dFn.apply(x, weight, bias, stride, pad, relu, want_stats,
                                 (param, slot, bslot) if param is not None else None, round_bias)
    if LAZY_BN_APPLY and want_stats and slot is not None and bias is None and not relu and stride == 1 and pad == 0 \
            and x.requires_grad and tuple(weight.shape[2:]) == (1, 1) and 128 < weight.shape[0] <= 256 and x.shape[3] <= 64 \
            and x.shape[0] * x.shape[1] * x.shape[2] >= FUSED_BWD_MIN_PIXELS:
        # a batch normalisation on this output may leave its backward apply step to this layer's fused backward launch
        out._u2_lazy_ok = True
    return (out, stats) if want_stats else out


def conv2d_add_(x, weight, bias, out, stride=1, pad=0, relu=False, param=None):
    """Inference only, in place: out = act(bf16(conv(x, weight) + bias) + out) - a residual block's tail (conv3 with its
    fixed-statistics norm folded in, + shortcut, + ReLU: backbone/resnet.py:204-210) as ONE launch that reads the shortcut
    where it writes the result.  `out` is overwritten: the caller must own it."""
    assert not torch.is_grad_enabled()
    _check_act(x)
    _check_act(out)
    n, cin, kh, kw = weight.shape
    b, h, w_, cp = x.shape
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (w_ + 2 * pad - kw) // stride + 1
    npad = ceil32(n)
    assert tuple(out.shape) == (b, ho, wo, npad) and npad == n, "the accumulating epilogue has no padded output channels"
    wk = weight_fwd_layout(weight, cp, param)
    bias_f = bias.detach().float().contiguous() if bias is not None else None
    _hip.call("u2_conv_igemm", x, wk, out, bias_f, None, b, h, w_, cp, cp, ho, wo, n, npad, kh, kw, pad, pad, stride, 1, int(relu),
              1, _conv_variant())
    return out


def linear(x2d, weight, bias=None, relu=False):
    """nn.Linear on a [R, K] bf16 matrix (K % 32 == 0); returns [R, ceil32(N)]."""
    r, k = x2d.shape
    n = weight.shape[0]
    param = weight if isinstance(weight, torch.nn.Parameter) else None
    out = conv2d(x2d.view(1, r, 1, k), weight.view(n, weight.shape[1], 1, 1), bias, 1, 0, relu, False, param)
    return out.view(r, -1)


class _StemConvFn(Function):
    """(x - mean)/std, zero pad to the batch canvas, 7x7 stride-2 pad-3 conv of 3 -> 64 channels as an
    im2col GEMM (rcnn.py:223-234 + backbone/resnet.py:355-357). Images need no gradient."""

    KP = 160

    @staticmethod
    def forward(ctx, weight, images, pixel_mean, pixel_std, hpad, wpad):
        n = weight.shape[0]
        b = len(images)
        ho, wo = (hpad + 6 - 7) // 2 + 1, (wpad + 6 - 7) // 2 + 1
        kp = _StemConvFn.KP
        import ctypes

        col = torch.empty((b * ho * wo, kp), dtype=BF16, device=weight.device)
        is_u8 = images[0].dtype == torch.uint8
        for img in images:
            assert img.is_cuda and img.is_contiguous() and img.shape[0] == 3
            assert img.dtype == (torch.uint8 if is_u8 else torch.float32), "one pixel type per batch"
        ptrs = (ctypes.c_void_p * b)(*[img.data_ptr() for img in images])
        hs = (ctypes.c_int * b)(*[img.shape[1] for img in images])
        ws = (ctypes.c_int * b)(*[img.shape[2] for img in images])
        _hip.call("u2_stem_im2col_batch", ptrs, hs, ws, b, int(is_u8), pixel_mean, pixel_std, col, hpad, wpad, kp)
        # [64, 3, 7, 7] -> K order (kh, kw, c), zero padded to kp: the forward layout of a 49-tap, 3-channel conv flattened
        wk = torch.zeros((n, 1, kp), dtype=BF16, device=weight.device)
        wk[:, 0, :147] = weight.detach().permute(0, 2, 3, 1).reshape(n, 147)
        out = torch.empty((b, ho, wo, n), dtype=BF16, device=weight.device)
        stats = _stats_buffer(n, weight.device)
        m = b * ho * wo
        _hip.call("u2_conv_igemm", col, wk, out, None, None if _DET_STATS else stats, 1, m, 1, kp, kp, m, 1, n, n, 1, 1, 0, 0, 1,
                  1, 0, 0, 0)
        if _DET_STATS:
            stats = _fixed_order_stats(out, n)
        ctx.save_for_backward(col)
        ctx.n = n
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dout, _ds):
        (col,) = ctx.saved_tensors
        n, kp = ctx.n, _StemConvFn.KP
        dout = dout.contiguous()
        m = col.shape[0]
        dwk = zeros_f32((n, 1, kp), col.device)
        _hip.call("u2_conv_wgrad", col, dout, dwk, 1, m, 1, kp, kp, m, 1, n, n, 1, 1, 0, 0, 1, 0)
        dw = dwk[:, 0, :147].view(n, 7, 7, 3).permute(0, 3, 1, 2)
        return dw, None, None, None, None, None


def stem_conv(weight, images, pixel_mean, pixel_std, hpad, wpad):
    return _StemConvFn.apply(weight, images, pixel_mean, pixel_std, hpad, wpad)


# --------------------------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------------------------
class _BatchNormActFn(Function):
    """Training-mode (Sync)BatchNorm (+residual)(+ReLU) on a conv output whose column statistics were
    produced by the conv epilogue (reference: nn.SyncBatchNorm chosen at layers/batch_norm.py:187,
    residual add + relu_ at backbone/resnet.py:204-210)."""

    @staticmethod
    def forward(ctx, y, stats, gamma, beta, running_mean, running_var, residual, relu, momentum, eps, grad_dst=None,
                twin=False, sync=True, res_up=False, lazy_ok=False):
        _check_act(y)
        b, h, w, c = y.shape
        m = b * h * w
        ctx.lazy_ok = lazy_ok
        world = _world() if sync else 1
        count, count_dev = float(m), None
        if world > 1:
            # nn.SyncBatchNorm (layers/batch_norm.py:187) gathers (mean, invstd, count) of every rank: the ranks pad their
            # batches to their own maximum image size, so the element counts differ.  One all-reduce of [sum | sumsq | count].
            packed = getattr(stats, "_u2_packed", None)
            if packed is None or packed.numel() != 2 * c + 1:  # statistics that did not come from a conv epilogue (tests)
                packed = torch.cat([stats.reshape(-1).float(), stats.new_zeros(1, dtype=torch.float32)])
            packed[2 * c :].fill_(float(m))  # an asynchronous fill: no synchronising constant upload per batch shape
            dist.all_reduce(packed)
            stats, count_dev = packed[: 2 * c].view(2, c), packed[2 * c :]
        mean = torch.empty(c, dtype=torch.float32, device=y.device)
        invstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
        out = torch.empty_like(y)
        bits = None
        if res_up:
            # FPN top-down step (backbone/fpn.py:153-155): the residual is the coarser level, nearest-upsampled on the fly
            assert residual is not None and not relu and tuple(residual.shape) == (b, h // 2, w // 2, c)
            _hip.call("u2_bn_finalize_fwd", stats, count, count_dev, gamma, beta, running_mean, running_var, momentum, eps, mean,
                      invstd, scale, shift, c)
            _hip.call("u2_affine_upadd", y, scale, shift, residual.contiguous(), out, b, h, w, c, 0)
        else:
            # a residual block's tail: backward needs only the sign of the activation - kept as one bit per element, so that
            # the reduce pass reads a sixteenth of the activation's bytes for it
            cpr = c // 8
            if relu and residual is not None and c % 8 == 0 and cpr <= 256 and 256 % cpr == 0:
                bits = torch.empty((m, cpr), dtype=torch.uint8, device=y.device)
            # finalize + apply in one launch (round 4): the coefficients are derived per thread from the column sums
            _hip.call("u2_bn_act_fused", y, stats, count, count_dev, gamma, beta, running_mean, running_var, momentum, eps, mean,
                      invstd, scale, shift, residual, out, m, c, c, int(relu), bits)
        # ReLU mask in backward: without a residual it is recomputed from y (one activation read less per pass)
        remask = relu and residual is None
        keep_out = relu and not remask
        ctx.mask_bits = bool(keep_out and bits is not None)
        ctx.save_for_backward(y, (bits if ctx.mask_bits else out) if keep_out else None, gamma, mean, invstd,
                              scale if remask else None, shift if remask else None)
        ctx.cfg = (relu, count, world, residual is not None)
        ctx.count_dev = count_dev
        ctx.res_up = res_up
        ctx.grad_dst = grad_dst  # (dgamma, dbeta) arena slices or None
        ctx.twin = twin
        if twin:  # two or three handles on the same activation: their gradients arrive separately and are summed in the kernel
            ctx.set_materialize_grads(False)
            return (out, out.detach(), out.detach()) if int(twin) == 3 else (out, out.detach())
        return out

    @staticmethod
    def backward(ctx, dout, dout2=None, dout3=None):
        y, out, gamma, mean, invstd, msc, msh = ctx.saved_tensors
        relu, count, world, has_res = ctx.cfg
        b, h, w, c = y.shape
        m = b * h * w
        nret = 15
        arrived = [g.contiguous() for g in (dout, dout2, dout3) if g is not None]
        if not arrived:
            return (None,) * nret
        fuse = has_res and relu  # residual block tail: dz is materialised by the reduce pass and is the residual gradient
        if not fuse and len(arrived) > 1:
            total = arrived[0]
            for g in arrived[1:]:
                total = total + g
            arrived = [total]
        dout, dout2, dout3 = (arrived + [None, None])[:3]
        dz = torch.empty_like(y) if fuse else None
        sums = zeros_f32((2, c), y.device)
        _hip.call("u2_norm_bwd_reduce", dout, out, y, mean, invstd, sums, 1, m, c, c, int(relu), msc, msh, dout2, dz, dout3,
                  int(bool(ctx.mask_bits and fuse)))
        local = sums
        if world > 1:
            local = sums.clone()
            dist.all_reduce(sums)
        coef = torch.empty((5, c), dtype=torch.float32, device=y.device)
        direct = ctx.grad_dst is not None
        dgamma, dbeta = ctx.grad_dst if direct else (coef[0], coef[1])
        if ctx.lazy_ok and LAZY_BN_APPLY and (fuse or (not relu and not has_res)):
            # the producing 1x1 conv evaluates the apply step dx = k1 dz + k2 y + k3 on its staged rows (u2_conv1x1_bwd_fused_bn,
            # round 5): only the coefficients (and dgamma / dbeta) are computed here, dx is never stored.  What autograd carries
            # to the conv is a one-element placeholder expanded to the shape; _Conv2dFn.backward finds the operands by its address.
            _hip.call("u2_bn_finalize_bwd", sums, count, ctx.count_dev, gamma, mean, invstd, local, dgamma, dbeta, coef[2], coef[3],
                      coef[4], c, int(direct))
            ph = _lazy_placeholder(y.device).expand(y.shape)
            _LAZY_GRADS[(ph.device.index, ph.data_ptr())] = (ph, dz if fuse else dout, y, coef)
            if isinstance(ctx.lazy_ok, dict):
                ctx.lazy_ok["key"] = (ph.device.index, ph.data_ptr())   # what _lazy_guard must see arrive at the conv output
            return ph, None, (None if direct else dgamma), (None if direct else dbeta), None, None, (dz if fuse else None), \
                None, None, None, None, None, None, None, None
        dx = torch.empty_like(y)
        # finalize (coefficients of dx, dgamma / dbeta) + apply in one launch (round 4); coef[2:5] is scratch for the channel
        # counts the one-launch form does not serve
        if fuse:
            _hip.call("u2_bn_bwd_apply_fused", sums, count, ctx.count_dev, gamma, mean, invstd, local, dgamma, dbeta, coef[2:],
                      int(direct), dz, None, y, dx, None, m, c, c, 0, None, None)
            dres = dz
        else:
            dres = torch.empty_like(y) if (has_res and not ctx.res_up) else None
            _hip.call("u2_bn_bwd_apply_fused", sums, count, ctx.count_dev, gamma, mean, invstd, local, dgamma, dbeta, coef[2:],
                      int(direct), dout, out, y, dx, dres, m, c, c, int(relu), msc, msh)
            if ctx.res_up:  # gradient of the coarser level: the 2x2 sums of dout
                dres = torch.empty((b, h // 2, w // 2, c), dtype=BF16, device=y.device)
                _hip.call("u2_fpn_upsample_add_bwd", dout, dres, b, h, w, c)
        if direct:
            dgamma = dbeta = None
        return dx, None, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None, None


def batch_norm_act(y, stats, gamma, beta, running_mean, running_var, residual=None, relu=False, momentum=0.1,
                   eps=1e-5, twin=False, sync=True, res_up=False):
    """twin=True: returns the activation with a second autograd handle on the same memory in `._u2_twin` (for a consumer
    pair such as the next residual block's conv1 and identity shortcut); the two gradients are summed inside the
    backward kernel instead of by autograd.  twin=3: a third handle in `._u2_third` as well.  sync=False: per-process statistics (NORM "BN"), no all-reduce.
    res_up=True: `residual` is the next coarser FPN level [B, H/2, W/2, C], nearest-upsampled inside the same pass."""
    gd, bd = grad_slot(gamma), grad_slot(beta)
    grad_dst = (gd, bd) if gd is not None and bd is not None else None
    twin = (3 if twin == 3 else int(bool(twin))) if torch.is_grad_enabled() else 0
    lazy_ok = bool(getattr(y, "_u2_lazy_ok", False)) and y.requires_grad and torch.is_grad_enabled()
    if lazy_ok:
        # the conv output must reach its convolution's backward as the placeholder itself: a second consumer of y, or anything
        # else that makes autograd form a sum, would silently drop the deferred gradient - fail there and then instead
        lazy_ok = {"key": None}
        y.register_hook(functools.partial(_lazy_guard, lazy_ok))
    out = _BatchNormActFn.apply(y, stats, gamma, beta, running_mean, running_var, residual, relu, momentum, eps, grad_dst,
                                twin, sync, res_up, lazy_ok)
    if twin == 3:  # a third handle (`._u2_third`) for a third consumer, e.g. the FPN lateral conv on a stage output
        out, other, third = out
        out._u2_twin, out._u2_third = other, third
    elif twin:
        out, other = out
        out._u2_twin = other
    return out


def affine_act(y, scale, shift, residual=None, relu=False):
    """Inference-mode normalisation: y*scale + shift (+res)(relu); no autograd."""
    b, h, w, c = y.shape
    out = torch.empty_like(y)
    _hip.call("u2_affine_act", y, scale.contiguous(), shift.contiguous(), residual, out, 1, b * h * w, c, c, int(relu), None)
    return out


class _GroupNormActFn(Function):
    """nn.GroupNorm(G, C) (+ReLU) on NHWC (reference: layers/batch_norm.py:189, semantic_seg.py:196-205)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, groups, relu, eps, grad_dst=None):
        _check_act(y)
        b, h, w, c = y.shape
        hw = h * w
        cg = c // groups
        ctx.grad_dst = grad_dst  # (dgamma, dbeta) arena slices or None
        stats = zeros_f32((b, 2, c), y.device)
        _hip.call("u2_colstats", y, stats, b, hw, c, c)
        coef = torch.empty((4, b, c), dtype=torch.float32, device=y.device)
        mean, invstd, scale, shift = coef[0], coef[1], coef[2], coef[3]
        _hip.call("u2_gn_finalize_fwd", stats, gamma.detach(), beta.detach(), float(hw * cg), float(eps), b, c, groups, mean,
                  invstd, scale, shift)
        out = torch.empty_like(y)
        _hip.call("u2_affine_act", y, scale, shift, None, out, b, hw, c, c, int(relu), None)
        ctx.save_for_backward(y, gamma, mean, invstd, scale if relu else None, shift if relu else None)
        ctx.cfg = (groups, relu)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, mean, invstd, msc, msh = ctx.saved_tensors
        groups, relu = ctx.cfg
        b, h, w, c = y.shape
        hw, cg = h * w, c // groups
        n = float(hw * cg)
        dout = dout.contiguous()
        sums = zeros_f32((b, 2, c), y.device)
        _hip.call("u2_norm_bwd_reduce", dout, None, y, mean, invstd, sums, b, hw, c, c, int(relu), msc, msh, None, None, None, 0)
        coef = torch.empty((3, b, c), dtype=torch.float32, device=y.device)
        k1, k2, k3 = coef[0], coef[1], coef[2]
        direct = ctx.grad_dst is not None
        if direct:   # accumulated into the optimizer's arena slices by the finalize kernel: nothing for AccumulateGrad to add
            dg, db = ctx.grad_dst
        else:
            dparam = torch.empty((2, c), dtype=torch.float32, device=y.device)
            dg, db = dparam[0], dparam[1]
        _hip.call("u2_gn_finalize_bwd", sums, gamma.detach(), mean, invstd, n, b, c, groups, k1, k2, k3, dg, db, int(direct))
        dx = torch.empty_like(y)
        _hip.call("u2_norm_bwd_apply", dout, None, y, k1, k2, k3, dx, None, b, hw, c, c, int(relu), msc, msh)
        return dx, (None if direct else dg), (None if direct else db), None, None, None, None


def group_norm_act(y, gamma, beta, groups, relu=False, eps=1e-5):
    gd, bd = grad_slot(gamma), grad_slot(beta)
    ok = gd is not None and bd is not None and gd.is_contiguous() and bd.is_contiguous() and gd.dtype == torch.float32
    return _GroupNormActFn.apply(y, gamma, beta, groups, relu, eps, (gd, bd) if ok else None)


# --------------------------------------------------------------------------------------------
# pooling / resampling
# --------------------------------------------------------------------------------------------
class _MaxPoolFn(Function):
    @staticmethod
    def forward(ctx, x):
        _check_act(x)
        b, h, w, c = x.shape
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        y = torch.empty((b, ho, wo, c), dtype=BF16, device=x.device)
        idx = torch.empty((b, ho, wo, c), dtype=torch.uint8, device=x.device)
        _hip.call("u2_maxpool3x3s2_fwd", x, y, idx, b, h, w, c)
        ctx.save_for_backward(idx)
        ctx.shape = (b, h, w, c)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        b, h, w, c = ctx.shape
        dx = torch.empty((b, h, w, c), dtype=BF16, device=dy.device)
        _hip.call("u2_maxpool3x3s2_bwd", dy.contiguous(), idx, dx, b, h, w, c)
        return dx


def max_pool_3x3_s2(x):
    return _MaxPoolFn.apply(x)


STEM_TAIL_FUSE = os.environ.get("U2_STEM_TAIL_FUSE", "1") != "0"


class _BatchNormReluMaxPoolFn(Function):
    """The stem's tail, norm -> relu_ -> max_pool2d(3, 2, 1) (backbone/resnet.py:355-359), without the activation between the
    normalisation and the pool (round 6): the pool normalises its nine taps itself, and the normalisation's backward passes
    rebuild the pool's gradient per input pixel from (dy, idx).  Pooled values and winner slots are those of
    `max_pool_3x3_s2(batch_norm_act(y, ..., relu=True))`, bit for bit; the statistics steps (all-reduce of [sum | sumsq | count]
    forward, of [sum dz | sum dz xhat] backward) are `_BatchNormActFn`'s."""

    @staticmethod
    def forward(ctx, y, stats, gamma, beta, running_mean, running_var, momentum, eps, grad_dst, sync):
        _check_act(y)
        b, h, w, c = y.shape
        m = b * h * w
        world = _world() if sync else 1
        count, count_dev = float(m), None
        if world > 1:
            packed = getattr(stats, "_u2_packed", None)
            if packed is None or packed.numel() != 2 * c + 1:
                packed = torch.cat([stats.reshape(-1).float(), stats.new_zeros(1, dtype=torch.float32)])
            packed[2 * c :].fill_(float(m))
            dist.all_reduce(packed)
            stats, count_dev = packed[: 2 * c].view(2, c), packed[2 * c :]
        coef = torch.empty((4, c), dtype=torch.float32, device=y.device)
        mean, invstd, scale, shift = coef[0], coef[1], coef[2], coef[3]
        _hip.call("u2_bn_finalize_fwd", stats, count, count_dev, gamma, beta, running_mean, running_var, momentum, eps, mean,
                  invstd, scale, shift, c)
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        out = torch.empty((b, ho, wo, c), dtype=BF16, device=y.device)
        idx = torch.empty((b, ho, wo, c), dtype=torch.uint8, device=y.device)
        _hip.call("u2_affine_relu_maxpool_fwd", y, scale, shift, out, idx, b, h, w, c)
        ctx.save_for_backward(y, idx, gamma, mean, invstd, scale, shift)
        ctx.cfg = (count, world)
        ctx.count_dev = count_dev
        ctx.grad_dst = grad_dst
        return out

    @staticmethod
    def backward(ctx, dy):
        y, idx, gamma, mean, invstd, msc, msh = ctx.saved_tensors
        count, world = ctx.cfg
        b, h, w, c = y.shape
        dy = dy.contiguous()
        sums = zeros_f32((2, c), y.device)
        _hip.call("u2_affine_relu_maxpool_bwd_reduce", dy, idx, y, mean, invstd, msc, msh, sums, b, h, w, c)
        local = sums
        if world > 1:
            local = sums.clone()
            dist.all_reduce(sums)
        coef = torch.empty((5, c), dtype=torch.float32, device=y.device)
        direct = ctx.grad_dst is not None
        dgamma, dbeta = ctx.grad_dst if direct else (coef[0], coef[1])
        _hip.call("u2_bn_finalize_bwd", sums, count, ctx.count_dev, gamma, mean, invstd, local, dgamma, dbeta, coef[2], coef[3],
                  coef[4], c, int(direct))
        dx = torch.empty_like(y)
        _hip.call("u2_affine_relu_maxpool_bwd_apply", dy, idx, y, coef[2], coef[3], coef[4], msc, msh, dx, b, h, w, c)
        return dx, None, (None if direct else dgamma), (None if direct else dbeta), None, None, None, None, None, None


def stem_tail_ok(c):
    """Channel counts the one-pass stem tail serves (a thread keeps one 8-channel chunk: C / 8 must divide 256)."""
    return STEM_TAIL_FUSE and c % 8 == 0 and 1 <= c // 8 <= 256 and 256 % (c // 8) == 0


def batch_norm_relu_max_pool(y, stats, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, sync=True):
    gd, bd = grad_slot(gamma), grad_slot(beta)
    grad_dst = (gd, bd) if gd is not None and bd is not None else None
    return _BatchNormReluMaxPoolFn.apply(y, stats, gamma, beta, running_mean, running_var, momentum, eps, grad_dst, sync)


def affine_relu_max_pool(y, scale, shift):
    """Inference: max_pool_3x3_s2(affine_act(y, scale, shift, relu=True)) in one pass; no autograd."""
    b, h, w, c = y.shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    out = torch.empty((b, ho, wo, c), dtype=BF16, device=y.device)
    idx = torch.empty((b, ho, wo, c), dtype=torch.uint8, device=y.device)
    _hip.call("u2_affine_relu_maxpool_fwd", y, scale.contiguous(), shift.contiguous(), out, idx, b, h, w, c)
    return out


class _UpsampleAddFn(Function):
    @staticmethod
    def forward(ctx, lateral, top):
        _check_act(lateral)
        b, h, w, c = lateral.shape
        assert top.shape == (b, h // 2, w // 2, c), (lateral.shape, top.shape)
        out = torch.empty_like(lateral)
        _hip.call("u2_fpn_upsample_add_fwd", lateral, top.contiguous(), out, b, h, w, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        b, h, w, c = dout.shape
        dtop = torch.empty((b, h // 2, w // 2, c), dtype=BF16, device=dout.device)
        _hip.call("u2_fpn_upsample_add_bwd", dout, dtop, b, h, w, c)
        return dout, dtop


def fpn_upsample_add(lateral, top):
    return _UpsampleAddFn.apply(lateral, top)


class _BilinearUp2Fn(Function):
    @staticmethod
    def forward(ctx, x, addend):
        _check_act(x)
        b, h, w, c = x.shape
        out = torch.empty((b, 2 * h, 2 * w, c), dtype=BF16, device=x.device)
        _hip.call("u2_bilinear_up2_fwd", x, addend, out, b, h, w, c)
        ctx.shape = (b, h, w, c)
        ctx.has_add = addend is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        b, h, w, c = ctx.shape
        dout = dout.contiguous()
        # The semantic head sums its per-level branches at the common stride (semantic_seg.py:246-253), the last x2 upsample of
        # every branch adding the running sum: the SAME gradient tensor arrives here once per branch (it is handed on unchanged
        # as the addend's gradient), and its transposed upsample is the same tensor each time - computed once, shared (round 5:
        # 0.43 -> 0.16 ms of bilinear2_bwd_kernel per step).  The entry holds `dout` itself, so its address cannot be reused.
        global _UP2_BWD_LAST
        key = (dout.device.index, dout.data_ptr(), dout._version, tuple(dout.shape), torch.cuda.current_stream(dout.device).cuda_stream)
        hit = _UP2_BWD_LAST
        if hit is not None and hit[0] == key:
            dx = hit[2]
        else:
            dx = torch.empty((b, h, w, c), dtype=BF16, device=dout.device)
            _hip.call("u2_bilinear_up2_bwd", dout, dx, b, h, w, c)
            if ctx.has_add:   # (the engine runs a whole branch before the next one's node: the other upsamples must not evict it)
                _UP2_BWD_LAST = (key, dout, dx)
        return dx, (dout if ctx.has_add else None)


_UP2_BWD_LAST = None


def bilinear_up2(x, addend=None):
    return _BilinearUp2Fn.apply(x, addend)


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------
class _SemSegLossFn(Function):
    """bilinear x4 + cross_entropy(mean, ignore) fused (meta_arch/semantic_seg.py:255-267)."""

    @staticmethod
    def forward(ctx, logits, target_u8, num_classes, ignore):
        _check_act(logits)
        b, h, w, lp = logits.shape
        assert target_u8.dtype == torch.uint8 and target_u8.shape == (b, 4 * h, 4 * w)
        acc = torch.empty((b, h, w, lp), dtype=torch.float32, device=logits.device)  # every element is written
        sc = zeros_f32((2,), logits.device)
        _hip.call("u2_semseg_upsample_ce", logits, target_u8.contiguous(), acc, sc[0:1], sc[1:2], b, h, w, lp,
                  num_classes, ignore)
        ctx.save_for_backward(acc, sc)
        return sc[0] / sc[1]

    @staticmethod
    def backward(ctx, g):
        acc, sc = ctx.saved_tensors
        d = torch.empty(acc.shape, dtype=BF16, device=acc.device)
        gg = g.detach().float().reshape(1).contiguous()
        _hip.call("u2_scale_to_bf16", acc, gg, sc[1:2], 1.0, d, acc.numel())
        return d, None, None, None


def sem_seg_loss(logits, target_u8, num_classes, ignore=255):
    return _SemSegLossFn.apply(logits, target_u8, num_classes, ignore)


class _SoftmaxCEFn(Function):
    """cross_entropy(scores, labels, reduction='mean') (roi_heads/fast_rcnn.py:344)."""

    @staticmethod
    def forward(ctx, logits, labels, num_classes):
        r, lp = logits.shape
        d = torch.empty_like(logits)
        loss = zeros_f32((1,), logits.device)
        _hip.call("u2_softmax_ce", logits.contiguous(), labels.contiguous(), d, loss, r, num_classes, lp,
                  1.0 / max(r, 1))
        ctx.save_for_backward(d)
        return loss[0] / max(r, 1)

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g, None, None   # g: 0-dim fp32, the product keeps d's dtype (one launch)


def softmax_cross_entropy(logits, labels, num_classes):
    return _SoftmaxCEFn.apply(logits, labels, num_classes)


class _BoxRegL1Fn(Function):
    """smooth_l1(beta=0) = L1 over foreground rows, summed, / normalizer (roi_heads/fast_rcnn.py:424-463)."""

    @staticmethod
    def forward(ctx, pred, proposals, gt_boxes, labels, bg_label, weights, normalizer):
        r, lp = pred.shape
        d = torch.empty_like(pred)
        loss = zeros_f32((1,), pred.device)
        _hip.call("u2_box_reg_l1", pred.contiguous(), proposals.contiguous(), gt_boxes.contiguous(),
                  labels.contiguous(), d, loss, r, lp, bg_label, weights[0], weights[1], weights[2], weights[3],
                  1.0 / normalizer)
        ctx.save_for_backward(d)
        return loss[0] / normalizer

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g, None, None, None, None, None, None


def box_reg_l1_loss(pred, proposals, gt_boxes, labels, bg_label, weights, normalizer):
    return _BoxRegL1Fn.apply(pred, proposals, gt_boxes, labels, bg_label, weights, float(normalizer))


class _MaskPredictBCEFn(Function):
    """1x1 predictor restricted to the gt-class channel + BCE-with-logits(mean)
    (roi_heads/mask_head.py:258 + :33-112)."""

    @staticmethod
    def forward(ctx, x, weight, bias, classes, target_u8, phased=False):
        n, ph, pw, c = x.shape
        if phased:  # x = the deconvolution's unshuffled phases [n, P, P, 4 C]: the kernel reads (and writes dx) in place
            ph, pw, c = 2 * ph, 2 * pw, c // 4
        p = ph * pw
        k = weight.shape[0]
        w2 = weight.detach().reshape(k, c).float().contiguous()
        b2 = bias.detach().float().contiguous()
        x, classes, target_u8 = x.contiguous(), classes.contiguous(), target_u8.contiguous()
        loss = zeros_f32((1,), x.device)
        denom = float(max(n * p, 1))
        # the loss alone here; the gradients come from a second launch in backward, scaled by the loss's upstream gradient inside
        # the kernel (forming dx here and multiplying it by that scalar in backward was a 0.2 ms pass over dx per step)
        _hip.call("u2_mask_predict_bce", x, w2, b2, classes, target_u8, None, None, None, loss, None, n, p, c, 1.0 / denom,
                  ph if phased else 0, None)
        ctx.save_for_backward(x, w2, b2, classes, target_u8)
        ctx.cfg = (n, p, c, k, denom, ph if phased else 0, weight.shape)
        return loss[0] / denom

    @staticmethod
    def backward(ctx, g):
        x, w2, b2, classes, target_u8 = ctx.saved_tensors
        n, p, c, k, denom, phased_side, wshape = ctx.cfg
        dx = torch.empty_like(x)
        dw, db = zeros_f32((k, c), x.device), zeros_f32((k,), x.device)
        _hip.call("u2_mask_predict_bce", x, w2, b2, classes, target_u8, dx, dw, db, None, None, n, p, c, 1.0 / denom, phased_side,
                  g.detach().float().reshape(1).contiguous())
        return dx, dw.view(wshape), db, None, None, None


def mask_predict_bce_loss(x, weight, bias, classes, target_u8, phased=False):
    return _MaskPredictBCEFn.apply(x, weight, bias, classes, target_u8, phased)


def mask_predict_prob(x, weight, bias, classes, phased=False):
    """mask_rcnn_inference (roi_heads/mask_head.py:115-158) with the 1x1 predictor folded in: fp32 [n, 1, 2P, 2P] probabilities of
    each detection's predicted class.  x: [n, 2P, 2P, 256] trunk output, or with `phased` the deconvolution's unshuffled GEMM
    output [n, P, P, 4 * 256] (ConvTranspose2d(..., shuffle=False)).  No autograd."""
    _check_act(x)
    n = x.shape[0]
    side = x.shape[1] * 2 if phased else x.shape[1]
    c = x.shape[3] // 4 if phased else x.shape[3]
    out = torch.empty((n, 1, side, side), dtype=torch.float32, device=x.device)
    if n:
        _hip.call("u2_mask_predict_prob", x.contiguous(), weight.detach().reshape(weight.shape[0], -1).contiguous(),
                  bias.detach().contiguous(), classes.to(torch.int64).contiguous(), out, n, side, c, int(phased))
    return out


class _RPNLossFn(Function):
    """RPN objectness BCE(sum) + localisation L1(sum) over all levels, both / normalizer
    (proposal_generator/rpn.py:366-429).  `fused`: each level is ONE map holding objectness (columns 0-2) and deltas (3-14)."""

    @staticmethod
    def forward(ctx, labels, match, gt_boxes, anchors_per_level, num_anchors, normalizer, fused, *obj_and_deltas):
        nl = len(anchors_per_level)
        objs, dlts = obj_and_deltas[:nl], obj_and_deltas[nl:]
        b, atot = labels.shape
        g = gt_boxes.shape[1]
        loss = zeros_f32((2,), labels.device)
        grads = []
        off = 0
        for lvl in range(nl):
            o = objs[lvl]
            hw = o.shape[1] * o.shape[2]
            if fused:
                go, d, gd = torch.empty_like(o), None, None
            else:
                d = dlts[lvl]
                go, gd = torch.empty_like(o), torch.empty_like(d)
            _hip.call("u2_rpn_loss_level", o, d, labels, match, gt_boxes, anchors_per_level[lvl], go, gd, loss, b, hw,
                      num_anchors, o.shape[3], d.shape[3] if d is not None else 0, atot, off, g, 1.0 / normalizer)
            grads.append((go, gd))
            off += hw * num_anchors
        assert off == atot
        ctx.grads = grads
        ctx.fused = fused
        ctx.num_anchors = num_anchors
        return loss[0] / normalizer, loss[1] / normalizer

    @staticmethod
    def backward(ctx, g_cls, g_loc):
        if ctx.fused:
            a = ctx.num_anchors
            go0 = ctx.grads[0][0]
            scale = torch.zeros(go0.shape[-1], dtype=torch.float32, device=go0.device)
            scale[:a] = g_cls
            scale[a : 5 * a] = g_loc
            scale = scale.to(go0.dtype)
            return (None, None, None, None, None, None, None, *[go * scale for go, _ in ctx.grads])
        gos = [go * g_cls for go, _ in ctx.grads]   # 0-dim fp32 factors: the products keep the maps' dtype
        gds = [gd * g_loc for _, gd in ctx.grads]
        return (None, None, None, None, None, None, None, *gos, *gds)


def rpn_losses(labels, match, gt_boxes, anchors_per_level, num_anchors, normalizer, objs, deltas):
    """objs / deltas: per level [B, H, W, LP] maps - or, when the head ran both predictors as one conv (the maps carry
    `_u2_rpn_fused`), the fused maps in `objs` (deltas are then views of them and are not used)."""
    if all(getattr(o, "_u2_rpn_fused", False) for o in objs):
        return _RPNLossFn.apply(labels, match, gt_boxes, anchors_per_level, num_anchors, float(normalizer), True, *objs)
    return _RPNLossFn.apply(labels, match, gt_boxes, anchors_per_level, num_anchors, float(normalizer), False, *objs, *deltas)


# --------------------------------------------------------------------------------------------
# ROI ops
# --------------------------------------------------------------------------------------------
def _level_arrays(feats, scales):
    import ctypes

    nl = len(feats)
    hs = (ctypes.c_int * nl)(*[f.shape[1] for f in feats])
    ws = (ctypes.c_int * nl)(*[f.shape[2] for f in feats])
    sc = (ctypes.c_float * nl)(*scales)
    return hs, ws, sc


def _roi_group(rois, levels, b, nl):
    """ROIs grouped by (image, level) for the per-pixel gather: order int32 [R], segment offsets int32 [b*nl+1]."""
    r = rois.shape[0]
    if b * nl <= 256 and r <= 32768 and 512 * (b * nl + 2) + 2 * r <= 155648 and levels.dtype == torch.int32:
        order = torch.empty(r, dtype=torch.int32, device=rois.device)
        seg = torch.empty(b * nl + 1, dtype=torch.int32, device=rois.device)
        _hip.call("u2_roi_group", rois.contiguous(), levels.contiguous(), order, seg, r, b, nl)
        return order, seg
    key = rois[:, 0].to(torch.int64) * nl + levels.to(torch.int64)
    order = torch.argsort(key, stable=True).to(torch.int32)
    seg = torch.zeros(b * nl + 1, dtype=torch.int32, device=rois.device)
    seg[1:] = torch.cumsum(torch.bincount(key, minlength=b * nl), 0)
    return order, seg


def _roi_gather(shapes, scales, sets, device, level=None, addends=()):
    """sets: [(rois, order, seg, dout, P, gscale)] (at most 4) -> per-level bf16 gradient maps, each written once.
    level = l: that level's map only (a one-element list), with up to two more bf16 gradient maps of its shape (`addends`: what
    the map's other readers produced) added in fp32 before the rounding (u2_roi_align_bwd_gather_sum)."""
    import ctypes

    nl, ns = len(shapes), len(sets)
    hs = (ctypes.c_int * nl)(*[s[1] for s in shapes])
    ws = (ctypes.c_int * nl)(*[s[2] for s in shapes])
    sc = (ctypes.c_float * nl)(*scales)
    lv = range(nl) if level is None else [level]
    gbuf = {l: torch.empty(shapes[l], dtype=BF16, device=device) for l in lv}
    ptrs = (ctypes.c_void_p * nl)(*[gbuf[l].data_ptr() if l in gbuf else None for l in range(nl)])
    keep = [tuple(t.contiguous() if isinstance(t, torch.Tensor) else t for t in st) for st in sets]
    arr = lambda k: (ctypes.c_void_p * ns)(*[st[k].data_ptr() for st in keep])
    ps = (ctypes.c_int * ns)(*[st[4] for st in keep])
    gs = (ctypes.c_float * ns)(*[float(st[5]) for st in keep])
    if level is None:
        _hip.call("u2_roi_align_bwd_gather_multi", ptrs, hs, ws, sc, nl, ns, arr(0), arr(1), arr(2), arr(3), ps, gs,
                  shapes[0][0], shapes[0][3])
        return [gbuf[l] for l in range(nl)]
    assert len(addends) <= 2 and all(tuple(a.shape) == tuple(shapes[level]) and a.dtype == BF16 and a.is_contiguous() for a in addends)
    adds = list(addends) + [None, None]
    a0 = (ctypes.c_void_p * nl)(*[adds[0].data_ptr() if (l == level and adds[0] is not None) else None for l in range(nl)])
    a1 = (ctypes.c_void_p * nl)(*[adds[1].data_ptr() if (l == level and adds[1] is not None) else None for l in range(nl)])
    _hip.call("u2_roi_align_bwd_gather_sum", ptrs, hs, ws, sc, nl, 1 << level, ns, arr(0), arr(1), arr(2), arr(3), ps, gs, a0, a1,
              shapes[0][0], shapes[0][3])
    return [gbuf[level]]


class _FanOutFn(Function):
    """k autograd handles on one tensor: their gradients arrive separately and are summed by one kernel (u2_add_n) instead of
    autograd's k - 1 accumulation adds."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.set_materialize_grads(False)
        return tuple(x.detach() for _ in range(k))

    @staticmethod
    def backward(ctx, *grads):
        lazy = None
        if _ROI_LAZY:   # one of the gradients may be the placeholder of a deferred ROIAlign gather (roi_grad_tap, round 6)
            for g in grads:
                if g is not None:
                    ent = _ROI_LAZY.pop((g.device.index, g.data_ptr()), None)
                    if ent is not None:
                        assert lazy is None
                        lazy, grads = ent, tuple(h for h in grads if h is not g)
        gs = [g.contiguous() for g in grads if g is not None]
        if not gs and lazy is None:
            return None, None
        while len(gs) > (2 if lazy is not None else 1):
            part, gs = gs[:4], gs[4:]
            if part[0].dtype != BF16 or part[0].numel() % 8:
                total = part[0]
                for g in part[1:]:
                    total = total + g
            else:
                total = torch.empty_like(part[0])
                part = part + [None] * (4 - len(part))
                _hip.call("u2_add_n", part[0], part[1], part[2], part[3], total, total.numel())
            gs.insert(0, total)
        if lazy is not None:
            # the ROI poolers' gradient of this map is formed HERE, with the other readers' gradients added inside the gather
            # (fp32, one rounding): the separate u2_add_n pass read the gathered map back and the other two again
            shared, level = lazy
            here = torch.cuda.current_stream(shared["device"])
            for entry in shared["pend"]:
                ev = entry[6] if len(entry) > 6 else None
                if ev is not None:
                    here.wait_event(ev)
                    for t in entry[:4]:
                        t.record_stream(here)
            if any(g.dtype != BF16 for g in gs):
                gs = [g.to(BF16) for g in gs]
            out = _roi_gather(shared["shapes"], shared["scales"], shared["pend"], shared["device"], level, gs)[0]
            return out, None
        return gs[0], None


def fan_out(x, k):
    """`k` handles on `x` for `k` consumers (training only; without autograd the tensor itself k times)."""
    if k <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * max(k, 1)
    outs = _FanOutFn.apply(x, k)
    for o in outs:
        o._u2_fan = True   # its gradient goes straight to _FanOutFn.backward, which can take a deferred ROI gather (roi_grad_tap)
    return outs


class RoiGradTap:
    """Shared state of one roi_grad_tap(): the ROIAlign calls made on the tapped maps leave their (rois, dout) here in
    backward, and the tap's own backward turns all of them into the maps' gradient with one gather pass."""

    def __init__(self, feats, scales_hint=None):
        self.shapes = [tuple(f.shape) for f in feats]
        self.pending = []
        self.scales = None
        self.ids = None
        self.defer = False   # every map is a fan_out handle: the gather is left to _FanOutFn.backward (see there)


class _RoiGradTapFn(Function):
    @staticmethod
    def forward(ctx, state, *feats):
        ctx.state = state
        ctx.set_materialize_grads(False)
        return tuple(f.detach() for f in feats)

    @staticmethod
    def backward(ctx, *gfeats):
        st = ctx.state
        grads = list(gfeats)
        pend, st.pending = st.pending, []
        if st.defer and ROI_SUM_FOLD and pend and len(pend) <= 4 and all(g is None for g in grads) and pend[0][3].is_cuda:
            # every tapped map is a fan_out handle: hand its _FanOutFn.backward a zero placeholder and let IT run the level's
            # gather, with the gradients of the map's other readers as addends (u2_roi_align_bwd_gather_sum)
            dev = pend[0][3].device
            shared = {"shapes": st.shapes, "scales": st.scales, "pend": pend, "device": dev}
            out = []
            for l, shape in enumerate(st.shapes):
                ph = _lazy_placeholder(dev, _ROI_LAZY).expand(shape)
                _ROI_LAZY[(dev.index, ph.data_ptr())] = (shared, l)
                out.append(ph)
            return (None, *out)
        here = torch.cuda.current_stream(pend[0][3].device) if pend and pend[0][3].is_cuda else None
        for entry in pend:  # a set left by a ROIAlign that ran on another stream (the mask head's): wait for it, keep it alive
            ev = entry[6] if len(entry) > 6 else None
            if ev is not None and here is not None:
                here.wait_event(ev)
                for t in entry[:4]:
                    t.record_stream(here)
        for i in range(0, len(pend), 4):
            gbuf = _roi_gather(st.shapes, st.scales, pend[i : i + 4], pend[i][3].device)
            grads = [g if old is None else old + g for old, g in zip(grads, gbuf)]
        return (None, *grads)


def roi_grad_tap(feats):
    """Identity on the FPN maps that defers the backward of every roi_align() made on its outputs: the cascade's three
    box poolers and the mask pooler (modeling/poolers.py:206-263 x 4) then cost one gather pass per level, and the four
    gradient maps per level that autograd would otherwise materialise and sum are never formed."""
    state = RoiGradTap(feats)
    state.defer = all(getattr(f, "_u2_fan", False) for f in feats)
    outs = _RoiGradTapFn.apply(state, *feats)
    state.ids = [id(o) for o in outs]
    for o in outs:
        o._u2_roi_tap = state
    return list(outs)


FUSED_BWD_MIN_PIXELS = int(os.environ.get("U2_FUSED_BWD_MIN_PIXELS", "200000"))
# the batch-norm backward apply step of an expanding 1x1 layer evaluated inside that layer's fused backward launch (0: separate)
LAZY_BN_APPLY = os.environ.get("U2_LAZY_BN_APPLY", "1") != "0"
_LAZY_GRADS = {}   # (device, placeholder address) -> (placeholder, dz, y, coefficients [5][C]: rows 2-4 = k1, k2, k3)


# the ROIAlign gather of a tapped FPN map deferred to the map's _FanOutFn.backward (round 6): (device, placeholder address) ->
# (shared state of the tap's backward, level); U2_ROI_SUM_FOLD=0: the tap gathers on the spot and u2_add_n sums afterwards
ROI_SUM_FOLD = os.environ.get("U2_ROI_SUM_FOLD", "1") != "0"
_ROI_LAZY = {}
_LAZY_POOL = {}    # device index -> [zero-filled bf16 pool, next slot]: placeholders are ZEROS, so that a sum autograd forms with
                   # one by accident is numerically the other addend (and is then caught by _lazy_guard / the checks below)
_LAZY_SLOTS = 64


def _lazy_placeholder(device, registry=None):
    registry = _LAZY_GRADS if registry is None else registry
    ent = _LAZY_POOL.get(device.index)
    if ent is None:
        ent = _LAZY_POOL[device.index] = [torch.zeros(_LAZY_SLOTS, dtype=BF16, device=device), 0]
    for _ in range(_LAZY_SLOTS):
        slot = ent[0][ent[1] : ent[1] + 1]
        ent[1] = (ent[1] + 1) % _LAZY_SLOTS
        key = (device.index, slot.data_ptr())
        if key not in _LAZY_GRADS and key not in _ROI_LAZY:
            return slot
    n = len(_LAZY_GRADS) + len(_ROI_LAZY)
    _LAZY_GRADS.clear()
    _ROI_LAZY.clear()
    raise RuntimeError("%d deferred gradients were not consumed by their consumers" % n)


def _lazy_guard(state, grad):
    """Tensor hook on a conv output whose batch normalisation defers its backward apply step: the gradient autograd delivers
    must be the placeholder that normalisation registered (`state["key"]`, None when it did not defer), untouched."""
    key, state["key"] = state["key"], None
    if key is not None and (grad.device.index, grad.data_ptr()) != key:
        pend = len(_LAZY_GRADS)
        _LAZY_GRADS.clear()
        raise RuntimeError("a convolution output with a deferred batch-norm gradient has a second consumer (autograd summed "
                           "the placeholder with another gradient); %d deferred gradients dropped - set U2_LAZY_BN_APPLY=0 "
                           "for graphs that re-use conv outputs" % pend)
    return None


def reset_deferred_gradients(strict=True):
    """Start of a training step (solver.FlatSGD.zero_grad): nothing deferred may be left over from an earlier backward pass -
    one that raised midway leaves entries behind, which must not meet the addresses of the next pass."""
    global _UP2_BWD_LAST
    _UP2_BWD_LAST = None
    if _LAZY_GRADS or _ROI_LAZY:
        n = len(_LAZY_GRADS) + len(_ROI_LAZY)
        _LAZY_GRADS.clear()
        _ROI_LAZY.clear()
        if strict:
            raise RuntimeError("%d deferred gradients of an earlier backward pass were never consumed" % n)


def assert_no_deferred_gradients():
    """After a backward pass: every deferred batch-norm gradient must have been consumed by its convolution."""
    global _UP2_BWD_LAST
    _UP2_BWD_LAST = None   # the shared upsample gradient of the semantic head: nothing of this pass stays referenced
    if _LAZY_GRADS or _ROI_LAZY:
        n, m = len(_LAZY_GRADS), len(_ROI_LAZY)
        _LAZY_GRADS.clear()
        _ROI_LAZY.clear()
        raise RuntimeError("%d deferred batch-norm gradients were not consumed by their convolutions, %d deferred ROIAlign "
                           "gathers not by their fan-out nodes" % (n, m))
ROI_ORDER_MIN = int(os.environ.get("U2_ROI_ORDER_MIN", "1000000000"))


def _roi_process_order(rois, levels, nimg, nlevels):
    """Processing order of the ROIAlign forward: sorted by (level, image, top row of the box at its level / 8)."""
    key = (levels.long() * nimg + rois[:, 0].long()) * 4096 + (rois[:, 2] * (0.25 / 8)).long().clamp_(0, 4095)
    return torch.argsort(key).to(torch.int32)


class _ROIAlignFn(Function):
    """Multi-level ROIAlign(aligned=True, sampling_ratio=0) (modeling/poolers.py:206-263)."""

    @staticmethod
    def forward(ctx, rois, levels, out_size, scales, grad_scale, tap, *feats):
        import ctypes

        nl = len(feats)
        for f in feats:
            _check_act(f)
        c = feats[0].shape[3]
        r = rois.shape[0]
        ptrs = (ctypes.c_void_p * nl)(*[f.data_ptr() for f in feats])
        hs, ws, sc = _level_arrays(feats, scales)
        out = torch.empty((r, out_size, out_size, c), dtype=BF16, device=rois.device)
        order = _roi_process_order(rois, levels, feats[0].shape[0], nl) if r >= ROI_ORDER_MIN else None
        _hip.call("u2_roi_align_fwd", ptrs, hs, ws, sc, nl, rois.contiguous(), levels.contiguous(), order, out, r, c, out_size,
                  out_size)
        ctx.save_for_backward(rois, levels)
        ctx.cfg = (out_size, scales, grad_scale, [tuple(f.shape) for f in feats])
        ctx.tap = tap
        return out

    @staticmethod
    def backward(ctx, dout):
        import ctypes

        rois, levels = ctx.saved_tensors
        out_size, scales, grad_scale, shapes = ctx.cfg
        nl = len(shapes)
        b = shapes[0][0]
        r, c = rois.shape[0], shapes[0][3]
        none = (None,) * 6
        if ROI_ALIGN_BWD_ATOMIC:
            hs = (ctypes.c_int * nl)(*[s[1] for s in shapes])
            ws = (ctypes.c_int * nl)(*[s[2] for s in shapes])
            sc = (ctypes.c_float * nl)(*scales)
            gbuf = [torch.zeros(s, dtype=torch.float32, device=dout.device) for s in shapes]
            ptrs = (ctypes.c_void_p * nl)(*[g.data_ptr() for g in gbuf])
            _hip.call("u2_roi_align_bwd", ptrs, hs, ws, sc, nl, rois.contiguous(), levels.contiguous(), dout.contiguous(),
                      r, c, out_size, out_size, float(grad_scale))
            return (*none, *[g.to(BF16) for g in gbuf])
        order, seg = _roi_group(rois, levels, b, nl)
        entry = (rois, order, seg, dout, out_size, grad_scale)
        tap = ctx.tap
        if tap is not None:  # deferred: the tap's backward gathers all pending sets at once
            tap.scales = scales
            if dout.is_cuda:  # the tap may run on another stream than this node: mark the point its inputs are complete at
                ev = torch.cuda.Event()
                ev.record()
                entry = entry + (ev,)
            tap.pending.append(entry)
            return (*none, *([None] * nl))
        return (*none, *_roi_gather(shapes, scales, [entry], dout.device))


ROI_ALIGN_BWD_ATOMIC = False  # True selects the fp32-atomic scatter variant (kept for A/B tests)


def roi_align(feats, rois, levels, out_size, scales, grad_scale=1.0):
    tap = getattr(feats[0], "_u2_roi_tap", None)
    if tap is not None and (ROI_ALIGN_BWD_ATOMIC or tap.ids != [id(f) for f in feats]):
        tap = None  # not exactly the tapped list of maps: gather on the spot
    return _ROIAlignFn.apply(rois, levels, out_size, tuple(scales), grad_scale, tap, *feats)


def assign_levels(boxes, min_level, max_level, canonical_size=224, canonical_level=4):
    n = boxes.shape[0]
    lv = torch.empty(n, dtype=torch.int32, device=boxes.device)
    _hip.call("u2_assign_levels", boxes.contiguous(), lv, n, min_level, max_level, float(canonical_size), canonical_level)
    return lv


def mask_crop(masks_u8, rois, size):
    """BitMasks.crop_and_resize (structures/masks.py:191-218): masks [Nm,H,W] uint8, rois [R,5]."""
    r = rois.shape[0]
    out = torch.empty((r, size, size), dtype=torch.uint8, device=rois.device)
    _hip.call("u2_mask_crop", masks_u8.contiguous(), rois.contiguous(), out, r, masks_u8.shape[1], masks_u8.shape[2], size)
    return out


def iou_match(boxes, gt, ngt, lo, hi, allow_low_quality):
    """boxes [n,4] (shared) or [B,n,4]; gt [B,G,4]; ngt [B] int32 -> match [B,n] int32, labels [B,n] int8."""
    per_image = boxes.dim() == 3
    b, g = gt.shape[0], gt.shape[1]
    n = boxes.shape[-2]
    dev = gt.device
    match = torch.empty((b, n), dtype=torch.int32, device=dev)
    mval = torch.empty((b, n), dtype=torch.float32, device=dev)
    labels = torch.empty((b, n), dtype=torch.int8, device=dev)
    gmax = torch.empty((b, g), dtype=torch.int32, device=dev) if allow_low_quality else None
    _hip.call("u2_iou_match", boxes.contiguous(), int(per_image), gt.contiguous(), ngt.contiguous(), match, mval, gmax,
              labels, b, n, g, float(lo), float(hi), int(allow_low_quality))
    return match, labels, mval


def sem_seg_upsample(x, num_classes, scale):
    """x [B, H, W, Cp] bf16 -> (fp32 [B, num_classes, H*scale, W*scale] bilinear-upsampled logits, int64 [B, H*scale, W*scale]
    argmax) (u2_semseg_upsample; inference only)."""
    _check_act(x)
    b, h, w, cp = x.shape
    out = torch.empty((b, num_classes, h * scale, w * scale), dtype=torch.float32, device=x.device)
    amax = torch.empty((b, h * scale, w * scale), dtype=torch.int64, device=x.device)
    _hip.call("u2_semseg_upsample", x, out, amax, b, h, w, cp, num_classes, int(scale))
    return out, amax


class _TopkSeg(ctypes.Structure):
    """U2TopkSeg of include/u2seg_hip.h."""
    _fields_ = ([(f, ctypes.c_void_p) for f in ("vals", "mask", "idx_in", "cnt_in", "out_vals", "out_idx", "out_cnt")]
                + [("row_stride", ctypes.c_longlong)]
                + [(f, ctypes.c_int) for f in ("dtype", "rows", "n", "group", "pitch", "mask_value", "k", "largest", "cnt_group",
                                               "idx_mod", "idx_mul", "reserved")])


def _ptr(t):
    return None if t is None else t.data_ptr()


def topk_rows(vals, k, largest=True, mask=None, mask_value=1, group=1, pitch=1, n=None, want_vals=True):
    """Row-wise selection with a total order (u2_topk_rows): the k best of every row ranked by (value descending if
    `largest` else ascending, index ascending).  vals: [rows, n] fp32 / bf16 contiguous, or - with group / pitch / n - the
    `group` valid columns of a [rows, n // group, pitch]-shaped map (element i at (i // group) * pitch + i % group).
    mask (int8 [rows, n]): only elements equal to mask_value take part.
    Returns (values fp32 [rows, k] or None, indices int32 [rows, k], counts int32 [rows])."""
    return topk_rows_multi([dict(vals=vals, k=k, largest=largest, mask=mask, mask_value=mask_value, group=group, pitch=pitch, n=n,
                                 want_vals=want_vals)])[0]


def topk_rows_multi(specs):
    """Several independent selections (each a dict of topk_rows' arguments) in two launches of u2_topk_rows_multi instead of
    one or two per selection: the per-level pre-NMS top-k of all feature levels, or the two draws of a sampler over the same keys.
    A row is streamed by ONE work-group (five to seven dependent sweeps): long rows are cut into equal segments that are ranked
    as rows of their own in the first launch (reporting positions in the full row), and the survivors (k per segment, the real
    ones counted) are ranked again in the second.  Returns a list of (values or None, indices, counts)."""
    first, second, results = [], [], [None] * len(specs)
    keep = []  # the intermediate tensors are referenced by raw pointers only: they must outlive the launches below
    for j, sp in enumerate(specs):
        vals, k, largest = sp["vals"], int(sp["k"]), int(bool(sp.get("largest", True)))
        if k > _TOPK_MAX_K:
            # beyond the kernel's LDS-resident survivor list (e.g. PRE_NMS_TOPK = 12000 summed over the levels): a stable
            # device sort gives the same total order
            results[j] = _topk_rows_sorted(sp)
            continue
        mask, mask_value = sp.get("mask"), int(sp.get("mask_value", 1))
        group, pitch, n, want_vals = int(sp.get("group", 1)), int(sp.get("pitch", 1)), sp.get("n"), sp.get("want_vals", True)
        assert vals.is_cuda and vals.dtype in (torch.float32, BF16) and vals.is_contiguous()
        dev, rows = vals.device, vals.shape[0]
        if n is None:
            assert vals.dim() == 2
            n = vals.shape[1]
        row_stride = vals[0].numel()
        if mask is not None:
            assert mask.dtype == torch.int8 and mask.is_contiguous() and tuple(mask.shape) == (rows, n)
        idx = torch.empty((rows, k), dtype=torch.int32, device=dev)
        out = torch.empty((rows, k), dtype=torch.float32, device=dev) if want_vals else None
        cnt = torch.empty((rows,), dtype=torch.int32, device=dev)
        results[j] = (out, idx, cnt)
        dtype = 1 if vals.dtype == BF16 else 0
        segs = _topk_segments(rows, n, group, pitch, row_stride, k)
        if segs > 1:
            seg_n = n // segs
            k1 = min(k, seg_n)
            v1 = torch.empty((rows * segs, k1), dtype=torch.float32, device=dev)
            i1 = torch.empty((rows * segs, k1), dtype=torch.int32, device=dev)
            c1 = torch.empty((rows * segs,), dtype=torch.int32, device=dev)
            keep.append((v1, i1, c1))
            first.append(_TopkSeg(_ptr(vals), _ptr(mask), None, None, _ptr(v1), _ptr(i1), _ptr(c1), row_stride // segs, dtype,
                                  rows * segs, seg_n, group, pitch, mask_value, k1, largest, 1, segs, seg_n, 0))
            # dtype 2: fp32 storage of bf16 values (the survivors of a bf16 map) - two digit sweeps instead of four
            second.append(_TopkSeg(_ptr(v1), None, _ptr(i1), _ptr(c1), _ptr(out), _ptr(idx), _ptr(cnt), segs * k1, 2 * dtype, rows,
                                   segs * k1, 1, 1, 0, k, largest, k1, 1, 0, 0))
        else:
            first.append(_TopkSeg(_ptr(vals), _ptr(mask), None, None, _ptr(out), _ptr(idx), _ptr(cnt), row_stride, dtype, rows, n,
                                  group, pitch, mask_value, k, largest, 1, 1, 0, 0))
    for batch in (first, second):
        batch.sort(key=lambda g: -g.n)  # the longest rows start first
        for at in range(0, len(batch), 8):
            part = batch[at:at + 8]
            _hip.call("u2_topk_rows_multi", (_TopkSeg * len(part))(*part), len(part))
    del keep
    return results


_TOPK_MAX_K = 16384  # u2_topk_rows keeps the k survivors of a row in LDS (8 bytes each)


def _topk_rows_sorted(sp):
    """topk_rows for k beyond the kernel's limit: torch's stable sort on the device under the same total order (value descending /
    ascending, index ascending; -0.0 equal to +0.0), same outputs (padding: index 0, value -/+ inf)."""
    vals, k, largest = sp["vals"], int(sp["k"]), bool(sp.get("largest", True))
    mask, mask_value = sp.get("mask"), int(sp.get("mask_value", 1))
    group, pitch, n = int(sp.get("group", 1)), int(sp.get("pitch", 1)), sp.get("n")
    rows = vals.shape[0]
    if n is None:
        n = vals.shape[1]
    v = vals.reshape(rows, -1)
    if group != 1 or pitch != 1:
        v = v.reshape(rows, -1, pitch)[:, : n // group, :group].reshape(rows, n)
    v = v[:, :n].float()
    fill = float("-inf") if largest else float("inf")
    part = torch.ones_like(v, dtype=torch.bool) if mask is None else (mask == mask_value)
    order = torch.sort(torch.where(part, v, torch.full_like(v, fill)), dim=1, descending=largest, stable=True)[1]
    # elements that do not take part may tie with real -inf / +inf values: rank them last explicitly
    order = torch.gather(order, 1, torch.sort((~torch.gather(part, 1, order)).to(torch.int8), dim=1, stable=True)[1])[:, :k]
    cnt = part.sum(dim=1).clamp(max=k).to(torch.int32)
    real = torch.arange(order.shape[1], device=v.device)[None] < cnt[:, None]
    idx = torch.where(real, order, torch.zeros_like(order)).to(torch.int32)
    out = None
    if sp.get("want_vals", True):
        out = torch.where(real, torch.gather(v, 1, order), torch.full_like(v[:, : order.shape[1]], fill))
    if idx.shape[1] < k:  # k larger than the row
        pad = k - idx.shape[1]
        idx = torch.nn.functional.pad(idx, (0, pad))
        out = torch.nn.functional.pad(out, (0, pad), value=fill) if out is not None else None
    return out, idx, cnt


def _topk_segments(rows, n, group, pitch, row_stride, k):
    """How many equal segments to cut the rows of a selection into (1 = none): only long rows that few work-groups would
    stream, a divisor of the group count (segments must tile the row exactly), segments still several times longer than k.
    The first launch streams n / s elements per work-group, the merging second one s * k (at about twice the cost per element:
    fp32 keys, carried indices): s is the admissible divisor nearest to sqrt(n / 2k)."""
    if n < 32768 or rows >= 128 or n % group or row_stride != (n // group) * pitch:
        return 1
    groups = n // group
    target = math.sqrt(n / (2.0 * k))
    best, best_d = 1, None
    for s in range(2, 33):
        if groups % s == 0 and n // s >= max(4096, 2 * k) and rows * s <= 256:
            d = abs(math.log(s / target))
            if best_d is None or d < best_d:
                best, best_d = s, d
    return best


def apply_deltas(src, deltas, weights, img_idx=None, sizes=None, clamp=math.log(1000.0 / 16)):
    n = src.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=src.device)
    do_clip = sizes is not None
    _hip.call("u2_apply_deltas", src.contiguous(), deltas.contiguous(), img_idx, sizes, out, n, weights[0], weights[1],
              weights[2], weights[3], float(clamp), int(do_clip))
    return out


class _RpnLevel(ctypes.Structure):
    """U2RpnLevel of include/u2seg_hip.h."""
    _fields_ = ([(f, ctypes.c_void_p) for f in ("deltas", "anchors", "idx", "scores")]
                + [(f, ctypes.c_int) for f in ("hwa", "k", "pitch", "reserved")])


def rpn_decode(levels, num_anchors, batch, kmax, sizes, weights, clamp, min_size):
    """Decoded, clipped and filtered RPN candidates of all levels in one launch (u2_rpn_decode).  levels: list of dicts with
    deltas (NHWC bf16 view whose channel 0 is the first delta), anchors [hwa, 4] fp32, idx [B, k] int32, scores [B, k] fp32.
    Returns boxes [B * L, kmax, 4], scores [B * L, kmax] (row = image * L + level), keep int8 [B * L, kmax], nonfinite int32 [1]."""
    dev = sizes.device
    rows = batch * len(levels)
    boxes = torch.empty((rows, kmax, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((rows, kmax), dtype=torch.float32, device=dev)
    keep = torch.empty((rows, kmax), dtype=torch.int8, device=dev)
    nonfinite = zeros_f32((1,), dev).view(torch.int32)
    segs = []
    for lv in levels:
        d, anc, idx, sc = lv["deltas"], lv["anchors"], lv["idx"], lv["scores"]
        assert d.dtype == BF16 and d.dim() == 4 and d.stride(3) == 1 and d.stride(2) == d.stride(1) // d.shape[2]
        assert anc.dtype == torch.float32 and anc.is_contiguous() and idx.dtype == torch.int32 and idx.is_contiguous()
        assert sc.dtype == torch.float32 and sc.is_contiguous() and tuple(idx.shape) == tuple(sc.shape) == (batch, idx.shape[1])
        hwa = d.shape[1] * d.shape[2] * num_anchors
        assert anc.shape[0] == hwa and d.stride(0) == d.shape[1] * d.shape[2] * d.stride(2)
        segs.append(_RpnLevel(d.data_ptr(), anc.data_ptr(), idx.data_ptr(), sc.data_ptr(), hwa, idx.shape[1], d.stride(2), 0))
    _hip.call("u2_rpn_decode", (_RpnLevel * len(segs))(*segs), len(segs), num_anchors, batch, kmax, sizes, float(weights[0]),
              float(weights[1]), float(weights[2]), float(weights[3]), float(clamp), float(min_size), boxes, scores, keep, nonfinite)
    return boxes, scores, keep, nonfinite


def batched_nms(boxes, group, counts, thr, max_keep):
    """boxes [B,n,4] sorted by descending score, group [B,n] int32, counts [B] int32 ->
    keep [B,max_keep] int32 (positions in the sorted order), nkeep [B] int32."""
    b, n = boxes.shape[0], boxes.shape[1]
    ws_bytes = _hip.call_nostream("u2_nms_workspace_bytes", b, n)
    ws = torch.empty(ws_bytes // 8 + 1, dtype=torch.int64, device=boxes.device)
    keep = torch.zeros((b, max_keep), dtype=torch.int32, device=boxes.device)
    nkeep = torch.zeros(b, dtype=torch.int32, device=boxes.device)
    _hip.call("u2_batched_nms", boxes.contiguous(), group.contiguous(), counts.contiguous(), ws, keep, nkeep, b, n,
              float(thr), max_keep)
    return keep, nkeep
