"""FPN semantic segmentation head (detectron2/modeling/meta_arch/semantic_seg.py:143-267)."""
import math

import torch
from torch import nn

from ..config import configurable
from ..layers import Conv2d, c2_msra_fill, get_norm
from ..layers import functional as F
from ..utils.registry import Registry

SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")


class _Upsample2(nn.Module):
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False); parameter-free placeholder that keeps
    the reference's nn.Sequential indices (p4.0 conv, p4.1 upsample, p4.2 conv, ...)."""

    def forward(self, x, addend=None):
        return F.bilinear_up2(x, addend)


@SEM_SEG_HEADS_REGISTRY.register()
class SemSegFPNHead(nn.Module):
    """One stack per FPN level that brings the level to the common stride - (3x3 conv + norm + ReLU, x2 bilinear) once per
    octave above it, a single conv at the common stride itself - their sum, and a 1x1 predictor
    (meta_arch/semantic_seg.py:143-267).  Child modules carry the reference's names (`p2.0`, `p4.2`, `predictor`, ...): the
    level stacks are registered under the feature names, upsampling steps keep their slot in the stack's numbering."""

    @configurable
    def __init__(self, input_shape, *, num_classes, conv_dims, common_stride, loss_weight=1.0, norm=None, ignore_value=-1):
        super().__init__()
        levels = sorted((spec.stride, name, spec.channels) for name, spec in input_shape.items())
        if not levels:
            raise ValueError("the semantic head needs at least one input feature map")
        self.num_classes, self.common_stride = num_classes, common_stride
        self.loss_weight, self.ignore_value = loss_weight, ignore_value
        self.in_features = [name for _, name, _ in levels]
        self.scale_heads = []
        for stride, name, channels in levels:
            stack = self._level_stack(stride, channels, conv_dims, common_stride, norm)
            self.add_module(name, stack)
            self.scale_heads.append(stack)
        self.predictor = Conv2d(conv_dims, num_classes, kernel_size=1)
        c2_msra_fill(self.predictor)

    @staticmethod
    def _level_stack(stride, channels, width, common_stride, norm):
        octaves = int(math.log2(stride) - math.log2(common_stride))
        ops, cin = [], channels
        for _ in range(max(1, octaves)):
            conv = Conv2d(cin, width, kernel_size=3, padding=1, bias=not norm, norm=get_norm(norm, width), activation="relu")
            c2_msra_fill(conv)
            ops.append(conv)
            if stride != common_stride:
                ops.append(_Upsample2())
            cin = width
        return nn.Sequential(*ops)

    @classmethod
    def from_config(cls, cfg, input_shape):
        head = cfg.MODEL.SEM_SEG_HEAD
        wanted = set(head.IN_FEATURES)
        return dict(input_shape={name: spec for name, spec in input_shape.items() if name in wanted},
                    num_classes=head.NUM_CLASSES, conv_dims=head.CONVS_DIM, common_stride=head.COMMON_STRIDE, norm=head.NORM,
                    loss_weight=head.LOSS_WEIGHT, ignore_value=head.IGNORE_VALUE)

    def layers(self, features):
        """Sum of the per-level heads at the common stride, then the 1x1 predictor (semantic_seg.py:246-253).
        The last upsample of each head adds the running sum in the same kernel."""
        x = None
        for f, head in zip(self.in_features, self.scale_heads):
            y = features[f]
            ops = list(head)
            for i, op in enumerate(ops):
                last = i == len(ops) - 1
                if isinstance(op, _Upsample2):
                    y = op(y, x if last else None)
                else:
                    y = op(y)
            if x is None or isinstance(ops[-1], _Upsample2):
                x = y
            else:
                x = x + y
        return self.predictor(x)

    def training_pieces(self, features, targets_u8, out):
        """The training forward pass as closures that launch it piece by piece, in the order of `layers` (level stacks from the
        common stride upwards, each adding the running sum; predictor + loss last): `out` receives {"loss_sem_seg"} with the last
        piece.  Same kernels in the same order as forward(): where the pieces are launched is the caller's business
        (layers.functional.defer_pieces)."""
        state = {"x": None}

        def level(f, head):
            def run():
                x, y = state["x"], features[f]
                ops = list(head)
                for i, op in enumerate(ops):
                    last = i == len(ops) - 1
                    y = op(y, x if last else None) if isinstance(op, _Upsample2) else op(y)
                state["x"] = y if x is None or isinstance(ops[-1], _Upsample2) else x + y
            return run

        def tail():
            loss = F.sem_seg_loss(self.predictor(state["x"]), targets_u8, self.num_classes, self.ignore_value)
            out["loss_sem_seg"] = loss * self.loss_weight
            state["x"] = None

        return [level(f, head) for f, head in zip(self.in_features, self.scale_heads)] + [tail]

    def forward(self, features, targets_u8=None):
        """targets_u8: [B, H, W] uint8 (ignore_value where unlabeled).  Training -> (None, {"loss_sem_seg"});
        inference -> ([B, num_classes, H, W] fp32 logits, {})."""
        x = self.layers(features)
        if self.training:
            loss = F.sem_seg_loss(x, targets_u8, self.num_classes, self.ignore_value)
            return None, {"loss_sem_seg": loss * self.loss_weight}
        # x4 bilinear (align_corners=False) of the fp32 logits and their argmax in one kernel; the argmax rides along on the
        # result tensor for PanopticFPN.inference (panoptic_fpn.py:173), which would otherwise read the logits back
        if self.num_classes > 64:  # the fused kernel keeps a pixel's class scores in registers (U2Seg configs: 28 classes)
            logits = torch.nn.functional.interpolate(x[..., : self.num_classes].permute(0, 3, 1, 2).float(),
                                                     scale_factor=self.common_stride, mode="bilinear", align_corners=False)
            logits.u2_argmax = logits.argmax(dim=1)
            return logits, {}
        logits, argmax = F.sem_seg_upsample(x, self.num_classes, self.common_stride)
        logits.u2_argmax = argmax
        return logits, {}


def build_sem_seg_head(cfg, input_shape):
    return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, input_shape)
