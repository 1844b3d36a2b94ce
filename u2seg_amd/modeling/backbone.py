"""ResNet-50 + FPN backbone with detectron2's module tree and registry entries
(detectron2/modeling/backbone/{backbone.py:11, resnet.py:100-694, fpn.py:17-268, build.py:7-33}).

Activations are NHWC bf16; the stem consumes the raw image list directly (normalisation, padding and the
7x7 stride-2 conv are one im2col GEMM)."""
import math

import torch
from torch import nn

from ..layers import Conv2d, ShapeSpec, c2_msra_fill, c2_xavier_fill, get_norm
from ..layers import functional as F
from ..utils.registry import Registry

BACKBONE_REGISTRY = Registry("BACKBONE")


class Backbone(nn.Module):
    def __init__(self):
        super().__init__()

    @property
    def size_divisibility(self):
        return 0

    @property
    def padding_constraints(self):
        return {}

    def output_shape(self):
        return {
            name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
            for name in self._out_features
        }


class BasicStem(nn.Module):
    """conv7x7 s2 (3 -> 64) + norm + ReLU + maxpool 3x3 s2 (resnet.py:330-359)."""

    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, 4
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=7, stride=2, padding=3, bias=False,
                            norm=get_norm(norm, out_channels))
        c2_msra_fill(self.conv1)

    def forward(self, images, pixel_mean, pixel_std, padded_hw):
        conv, norm = self.conv1, self.conv1.norm
        y, stats = F.stem_conv(conv.weight, images, pixel_mean, pixel_std, padded_hw[0], padded_hw[1])
        fuse = F.stem_tail_ok(y.shape[-1])   # norm -> relu -> pool as one pass each way (round 6)
        if self.training and hasattr(norm, "momentum"):
            norm.count_batch()
            if fuse:
                return F.batch_norm_relu_max_pool(y, stats, norm.weight, norm.bias, norm.running_mean, norm.running_var,
                                                  norm.momentum, norm.eps, sync=norm.sync)
            x = F.batch_norm_act(y, stats, norm.weight, norm.bias, norm.running_mean, norm.running_var, None, True,
                                 norm.momentum, norm.eps, sync=norm.sync)
        else:
            scale, shift = norm.eval_scale_shift()
            if fuse and not (torch.is_grad_enabled() and y.requires_grad):
                return F.affine_relu_max_pool(y, scale.float(), shift.float())
            x = F.affine_act(y, scale.float(), shift.float(), None, True)
        return F.max_pool_3x3_s2(x)


class BottleneckBlock(nn.Module):
    """1x1 -> 3x3 -> 1x1 with optional projection shortcut (resnet.py:100-210)."""

    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1):
        super().__init__()
        assert num_groups == 1 and dilation == 1, "grouped / dilated bottlenecks are not used by the U2Seg configs"
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.third_handle = False  # set by ResNet on the last block of a stage that is also a backbone output
        self.first_in_stage = True  # ResNet clears it on every block but the first of its stage
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        stride_1x1, stride_3x3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=stride_1x1, bias=False,
                            norm=get_norm(norm, bottleneck_channels), activation="relu")
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride_3x3, padding=1,
                            bias=False, norm=get_norm(norm, bottleneck_channels), activation="relu")
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False,
                            norm=get_norm(norm, out_channels))
        for layer in [self.conv1, self.conv2, self.conv3, self.shortcut]:
            if layer is not None:
                c2_msra_fill(layer)

    def forward(self, x):
        # the block input feeds conv1 and the shortcut; when the previous block handed out two autograd handles of it
        # (functional.batch_norm_act(twin=True)) the two gradients meet inside that block's BN backward kernel
        x_sc = getattr(x, "_u2_twin", x)
        out = self.conv1(x)
        out = self.conv2(out)
        shortcut = self.shortcut(x_sc) if self.shortcut is not None else x_sc
        # out += shortcut; relu_.  The last block of a stage that is also a backbone output hands out a third handle
        # at inference the shortcut buffer is given up: a projection shortcut is this block's own tensor, an identity shortcut
        # is the previous block's output, which inside a stage has no reader but this block
        owned = not torch.is_grad_enabled() and not self.training and (self.shortcut is not None or not self.first_in_stage)
        return self.conv3(out, residual=shortcut, relu=True, twin=3 if self.third_handle else True, residual_owned=owned)


class ResNet(Backbone):
    """stem + res2..res5 (resnet.py:362-458).  The feature table - one (name, stride, channels) row for the stem and every
    stage - is built first; strides, channels and the default output are read from it."""

    def __init__(self, stem, stages, out_features=None, freeze_at=0):
        super().__init__()
        if freeze_at != 0:
            raise NotImplementedError("BACKBONE.FREEZE_AT > 0 is not used by the U2Seg configs")
        self.stem = stem
        table = [("stem", stem.stride, stem.out_channels)]
        self.stages = []
        for number, blocks in enumerate(stages, start=2):
            seq = nn.Sequential(*blocks)
            self.add_module("res%d" % number, seq)  # state-dict prefix of the stage
            self.stages.append(seq)
            table.append(("res%d" % number, table[-1][1] * math.prod(blk.stride for blk in blocks), blocks[-1].out_channels))
        self.stage_names = tuple(row[0] for row in table[1:])
        self._out_feature_strides = {name: stride for name, stride, _ in table}
        self._out_feature_channels = {name: channels for name, _, channels in table}
        self._out_features = list(out_features) if out_features is not None else [table[-1][0]]
        # autograd-handle bookkeeping of this implementation (see BottleneckBlock.forward): which block opens its stage, and
        # which stage outputs are read by the next stage AND by the caller
        for seq in self.stages:
            for blk in list(seq)[1:]:
                if hasattr(blk, "first_in_stage"):
                    blk.first_in_stage = False
        for name, seq in zip(self.stage_names[:-1], self.stages[:-1]):
            if name in self._out_features and hasattr(seq[-1], "third_handle"):
                seq[-1].third_handle = True

    def forward(self, images, pixel_mean, pixel_std, padded_hw):
        outputs = {}
        x = self.stem(images, pixel_mean, pixel_std, padded_hw)
        for name, stage in zip(self.stage_names, self.stages):
            x = stage(x)
            if name in self._out_features:
                # a stage output read by the next stage (conv1 + shortcut: the two twin handles) AND by the caller: the caller
                # gets the third handle, so that all three gradients meet inside one BatchNorm backward kernel
                outputs[name] = getattr(x, "_u2_third", x)
        return outputs

    @staticmethod
    def make_stage(block_class, num_blocks, *, in_channels, out_channels, stride_per_block, **kwargs):
        blocks = []
        for i in range(num_blocks):
            blocks.append(block_class(in_channels=in_channels, out_channels=out_channels, stride=stride_per_block[i],
                                      **kwargs))
            in_channels = out_channels
        return blocks


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape):
    """resnet.py:613-694 for the bottleneck depths."""
    r = cfg.MODEL.RESNETS
    norm = r.NORM
    stem = BasicStem(in_channels=input_shape.channels, out_channels=r.STEM_OUT_CHANNELS, norm=norm)
    depth = r.DEPTH
    blocks_per_stage = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
    bottleneck_channels = r.NUM_GROUPS * r.WIDTH_PER_GROUP
    in_channels, out_channels = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS
    assert r.RES5_DILATION == 1 and not any(r.DEFORM_ON_PER_STAGE)
    stages = []
    for idx in range(4):
        first_stride = 1 if idx == 0 else 2
        stages.append(ResNet.make_stage(
            BottleneckBlock, blocks_per_stage[idx], in_channels=in_channels, out_channels=out_channels,
            stride_per_block=[first_stride] + [1] * (blocks_per_stage[idx] - 1),
            bottleneck_channels=bottleneck_channels, stride_in_1x1=r.STRIDE_IN_1X1, num_groups=r.NUM_GROUPS, norm=norm))
        in_channels = out_channels
        out_channels *= 2
        bottleneck_channels *= 2
    return ResNet(stem, stages, out_features=r.OUT_FEATURES, freeze_at=cfg.MODEL.BACKBONE.FREEZE_AT)


class _Subsample2Fn(torch.autograd.Function):
    """x[:, ::2, ::2, :] as a contiguous tensor; the gradient map is written by one kernel (u2_subsample2_bwd; autograd's own chain
    for the two slices and the copy is two zero fills and two strided copies)."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return x[:, ::2, ::2, :].contiguous()

    @staticmethod
    def backward(ctx, g):
        b, h, w, c = ctx.shape
        if g.is_cuda and g.dtype == torch.bfloat16 and c % 8 == 0:
            from .. import _hip

            dx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device)
            _hip.call("u2_subsample2_bwd", g.contiguous(), dx, b, h, w, c)
            return dx
        dx = g.new_zeros(ctx.shape)
        dx[:, ::2, ::2, :] = g
        return dx


class LastLevelMaxPool(nn.Module):
    """p6 = max_pool2d(p5, kernel 1, stride 2) = a stride-2 subsample (fpn.py:188-200)."""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward(self, x):
        return [_Subsample2Fn.apply(x) if x.requires_grad else x[:, ::2, ::2, :].contiguous()]


class FPN(Backbone):
    """Top-down pyramid over the bottom-up features (fpn.py:17-167).  A level table (level number = log2 stride, source feature,
    its channels) is built first; the conv pairs are registered from it under the reference's state-dict names
    fpn_lateral<level> / fpn_output<level> (finest level first, the bottom-up network last: the reference's key order)."""

    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        if fuse_type != "sum":
            raise NotImplementedError("FPN.FUSE_TYPE %r: the U2Seg configs sum the two paths" % (fuse_type,))
        shapes = bottom_up.output_shape()
        levels = [(int(math.log2(shapes[f].stride)), f, shapes[f].channels) for f in in_features]
        biased = norm == ""
        pairs = {}
        for level, _feature, channels in levels:
            lateral = Conv2d(channels, out_channels, kernel_size=1, bias=biased, norm=get_norm(norm, out_channels))
            output = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=biased,
                            norm=get_norm(norm, out_channels))
            for conv in (lateral, output):
                c2_xavier_fill(conv)
            self.add_module("fpn_lateral%d" % level, lateral)
            self.add_module("fpn_output%d" % level, output)
            pairs[level] = (lateral, output)
        # the order the pyramid is computed in: coarsest level first
        self.top_down = [(level, feature) + pairs[level] for level, feature, _ in reversed(levels)]
        self.top_block = top_block
        self.in_features = tuple(in_features)
        self.bottom_up = bottom_up
        out_levels = [level for level, _, _ in levels]
        if top_block is not None:
            out_levels += [out_levels[-1] + 1 + extra for extra in range(top_block.num_levels)]
        self._out_features = ["p%d" % level for level in out_levels]
        self._out_feature_strides = {"p%d" % level: 2 ** level for level in out_levels}
        self._out_feature_channels = dict.fromkeys(self._out_features, out_channels)
        self._size_divisibility = shapes[in_features[-1]].stride

    @property
    def size_divisibility(self):
        return self._size_divisibility

    @property
    def padding_constraints(self):
        return {"square_size": 0}

    def forward(self, images, pixel_mean, pixel_std, padded_hw):
        """fpn.py:126-167: lateral 1x1, nearest x2 of the coarser level + add, output 3x3."""
        with F.streamk_region():   # no branch stream has work in flight while these layers run, forward or backward (layers/functional.py)
            bottom_up_features = self.bottom_up(images, pixel_mean, pixel_std, padded_hw)
        return self.forward_features(bottom_up_features)

    def forward_features(self, bottom_up_features):
        maps, carry = {}, None
        for level, feature, lateral, output in self.top_down:
            # lateral 1x1 + its norm (+ nearest x2 of the coarser level + add: one pass over the map, fpn.py:141-158)
            src = bottom_up_features[feature]
            carry = lateral(src) if carry is None else lateral(src, residual_up=carry)
            maps["p%d" % level] = output(carry)
        ordered = [maps[name] for name in self._out_features if name in maps]
        if self.top_block is not None:
            ordered += list(self.top_block(maps[self.top_block.in_feature]))
        if len(ordered) != len(self._out_features):
            raise RuntimeError("FPN produced %d maps for %d declared outputs" % (len(ordered), len(self._out_features)))
        return dict(zip(self._out_features, ordered))


@BACKBONE_REGISTRY.register()
def build_resnet_fpn_backbone(cfg, input_shape):
    bottom_up = build_resnet_backbone(cfg, input_shape)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, top_block=LastLevelMaxPool(), fuse_type=cfg.MODEL.FPN.FUSE_TYPE)


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)
    assert isinstance(backbone, Backbone)
    return backbone
