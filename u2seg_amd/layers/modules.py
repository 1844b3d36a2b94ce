"""nn.Module layer wrappers with detectron2's names and state-dict keys, computing through the HIP
functions of layers/functional.py on NHWC bf16 activations.

Mirrors detectron2/layers/wrappers.py:87-139 (Conv2d = conv -> norm -> activation), layers/batch_norm.py:13-197
(FrozenBatchNorm2d, get_norm) and layers/shape_spec.py:8.
"""
from collections import namedtuple

import os

import torch
from torch import nn

from . import functional as F

ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=[None, None, None, None])


def c2_xavier_fill(module):
    """fvcore.nn.weight_init.c2_xavier_fill: kaiming_uniform_(a=1), zero bias."""
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def c2_msra_fill(module):
    """fvcore.nn.weight_init.c2_msra_fill: kaiming_normal_(fan_out, relu), zero bias."""
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def to_nhwc(x_nchw, pad_to=32):
    """logical NCHW float tensor -> physical NHWC bf16 with channels zero padded to a multiple of 32."""
    b, c, h, w = x_nchw.shape
    cp = (c + pad_to - 1) // pad_to * pad_to
    out = torch.zeros((b, h, w, cp), dtype=torch.bfloat16, device=x_nchw.device)
    out[..., :c] = x_nchw.permute(0, 2, 3, 1)
    return out


def to_nchw(x_nhwc, channels=None):
    c = channels if channels is not None else x_nhwc.shape[3]
    return x_nhwc[..., :c].permute(0, 3, 1, 2)


class BatchNorm2d(nn.Module):
    """Training-mode batch norm with cross-rank statistics when torch.distributed is initialised
    (the role of nn.SyncBatchNorm / nn.BatchNorm2d picked by get_norm, layers/batch_norm.py:182-189)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, sync=True):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.sync = sync  # False: NORM "BN" - statistics of this process only
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending_batches = 0  # folded into the buffer when it is read (state_dict): no per-step device op
        self._updates = 0  # training-mode forwards so far: the running statistics change through a raw-pointer kernel

    def count_batch(self):
        self._pending_batches += 1
        self._updates += 1

    def _flush_batches(self):
        if self._pending_batches:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_batches()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        super()._load_from_state_dict(*args, **kwargs)

    def eval_scale_shift(self):
        scale = self.weight.detach() * torch.rsqrt(self.running_var + self.eps)
        return scale, self.bias.detach() - self.running_mean * scale


class FrozenBatchNorm2d(nn.Module):
    """layers/batch_norm.py:13-120: fixed statistics and affine (buffers, not parameters)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def eval_scale_shift(self):
        scale = self.weight * torch.rsqrt(self.running_var + self.eps)
        return scale, self.bias - self.running_mean * scale


class GroupNorm(nn.Module):
    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))


def get_norm(norm, out_channels):
    """layers/batch_norm.py:169-197."""
    if norm is None:
        return None
    if isinstance(norm, str):
        if len(norm) == 0:
            return None
        norm = {
            "BN": lambda channels: BatchNorm2d(channels, sync=False),   # per-GPU statistics (layers/batch_norm.py:181)
            "SyncBN": BatchNorm2d,                                       # statistics over all ranks (:187)
            "FrozenBN": FrozenBatchNorm2d,
            "GN": lambda channels: GroupNorm(32, channels),
        }[norm]
    return norm(out_channels)


class Conv2d(nn.Module):
    """conv -> norm -> activation (layers/wrappers.py:87-134), fused the MI355X way: the conv epilogue produces
    the batch statistics, one elementwise pass applies normalisation (+ residual) (+ ReLU)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, norm=None,
                 activation=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.stride, self.padding = stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.norm = norm
        assert activation in (None, "relu"), "only ReLU is fused (the U2Seg graph uses nothing else)"
        self.activation = activation

    def forward(self, x, residual=None, relu=None, twin=False, residual_up=None, residual_owned=False):
        """residual_up: the coarser FPN level to add nearest-upsampled (fused into the BatchNorm apply in training).
        residual_owned: at inference the caller gives `residual` up - the result is written over it (one launch)."""
        if residual_up is not None and not (isinstance(self.norm, BatchNorm2d) and self.training):
            assert residual is None
            return F.fpn_upsample_add(self.forward(x, None, relu, twin), residual_up)
        relu = (self.activation == "relu") if relu is None else relu
        norm = self.norm
        if norm is None:
            assert residual is None
            return F.conv2d(x, self.weight, self.bias, self.stride, self.padding, relu=relu)
        assert self.bias is None, "a conv followed by a norm layer carries no bias in this graph"
        if isinstance(norm, BatchNorm2d) and self.training:
            y, stats = F.conv2d(x, self.weight, None, self.stride, self.padding, relu=False, want_stats=True)
            norm.count_batch()
            if residual_up is not None:
                assert residual is None
                return F.batch_norm_act(y, stats, norm.weight, norm.bias, norm.running_mean, norm.running_var, residual_up,
                                        relu, norm.momentum, norm.eps, twin=twin, sync=norm.sync, res_up=True)
            return F.batch_norm_act(y, stats, norm.weight, norm.bias, norm.running_mean, norm.running_var, residual,
                                    relu, norm.momentum, norm.eps, twin=twin, sync=norm.sync)
        if isinstance(norm, GroupNorm):
            assert residual is None
            y = F.conv2d(x, self.weight, None, self.stride, self.padding, relu=False)
            return F.group_norm_act(y, norm.weight, norm.bias, norm.num_groups, relu, norm.eps)
        # fixed statistics (eval-mode BN / FrozenBN): y * scale + shift is an affine map per output channel
        if not torch.is_grad_enabled():
            folded = self._folded_eval()
            if residual is None:
                # folded into the conv: weights scaled per output channel, shift as the bias, ReLU in the epilogue - no
                # elementwise pass at all (the conv kernels add a bias and clamp before the one bf16 rounding)
                return F.conv2d(x, folded[0], folded[1], self.stride, self.padding, relu=relu, param=folded[0], round_bias=False)
            if residual_owned and self.out_channels % 32 == 0 and os.environ.get("U2_EVAL_RESIDUAL_FUSE", "1") != "0":
                return F.conv2d_add_(x, folded[0], folded[1], residual, self.stride, self.padding, relu=relu, param=folded[0])
            y = F.conv2d(x, self.weight, None, self.stride, self.padding, relu=False)
            return F.affine_act(y, folded[2], folded[3], residual, relu)
        y = F.conv2d(x, self.weight, None, self.stride, self.padding, relu=False)
        scale, shift = norm.eval_scale_shift()
        return F.affine_act(y, scale.float(), shift.float(), residual, relu)

    def _folded_eval(self):
        """(scaled weight, shift, scale, shift) of conv + fixed-statistics norm, cached until any of the tensors involved
        changes through torch (version counters) or is re-allocated."""
        norm = self.norm
        parts = (self.weight, norm.weight, norm.bias, norm.running_mean, norm.running_var)
        # the optimizer (u2_sgd_clip_step) and the BN finalize kernel write through raw pointers and never bump `_version`:
        # the optimizer's step stamp and the norm's update counter are part of the key
        def stamp(t):
            st = getattr(t, "_u2_stamp", None)
            return st[0] if st is not None else 0

        key = tuple((t._version, t.data_ptr(), stamp(t)) for t in parts) + (getattr(norm, "_updates", 0),)
        hit = self.__dict__.get("_u2_fold")
        if hit is not None and hit[0] == key:
            return hit[1]
        with torch.no_grad():
            scale, shift = norm.eval_scale_shift()
            scale, shift = scale.float().contiguous(), shift.float().contiguous()
            wf = (self.weight.detach().float() * scale.view(-1, 1, 1, 1)).contiguous()
        val = (wf, shift, scale, shift)
        self.__dict__["_u2_fold"] = (key, val)
        return val


class Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward(self, x2d, relu=False):
        return F.linear(x2d, self.weight, self.bias, relu)


class ConvTranspose2d(nn.Module):
    """ConvTranspose2d(k=2, s=2) as one GEMM producing the four (dy, dx) phases, then a pixel shuffle
    (roi_heads/mask_head.py:256-258)."""

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2, padding=0):
        super().__init__()
        assert kernel_size == 2 and stride == 2 and padding == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 2, 2))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward(self, x, relu=False, shuffle=True):
        """shuffle=False: the GEMM output [b, h, w, (dy, dx, co)] as it stands, for a consumer that works per output pixel and
        reads phase (dy, dx) of source pixel (h, w) itself (functional.mask_predict_prob at inference)."""
        b, h, w, _ = x.shape
        co = self.out_channels
        w4 = self.weight.permute(2, 3, 1, 0).reshape(4 * co, self.in_channels, 1, 1)
        y = F.conv2d(x, w4, self.bias.repeat(4), 1, 0, relu=relu)  # [b, h, w, (dy, dx, co)]
        if not shuffle:
            return y
        y = y.view(b, h, w, 2, 2, co).permute(0, 1, 3, 2, 4, 5).reshape(b, 2 * h, 2 * w, co)
        return y.contiguous()
