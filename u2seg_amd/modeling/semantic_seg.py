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
    @configurable
    def __init__(self, input_shape, *, num_classes, conv_dims, common_stride, loss_weight=1.0, norm=None, ignore_value=-1):
        super().__init__()
        input_shape = sorted(input_shape.items(), key=lambda x: x[1].stride)
        if not len(input_shape):
            raise ValueError("SemSegFPNHead(input_shape=) cannot be empty!")
        self.in_features = [k for k, v in input_shape]
        feature_strides = [v.stride for k, v in input_shape]
        feature_channels = [v.channels for k, v in input_shape]
        self.ignore_value, self.common_stride, self.loss_weight = ignore_value, common_stride, loss_weight
        self.num_classes = num_classes
        self.scale_heads = []
        for in_feature, stride, channels in zip(self.in_features, feature_strides, feature_channels):
            head_ops = []
            head_length = max(1, int(math.log2(stride) - math.log2(self.common_stride)))
            for k in range(head_length):
                norm_module = get_norm(norm, conv_dims)
                conv = Conv2d(channels if k == 0 else conv_dims, conv_dims, kernel_size=3, stride=1, padding=1,
                              bias=not norm, norm=norm_module, activation="relu")
                c2_msra_fill(conv)
                head_ops.append(conv)
                if stride != self.common_stride:
                    head_ops.append(_Upsample2())
            self.scale_heads.append(nn.Sequential(*head_ops))
            self.add_module(in_feature, self.scale_heads[-1])
        self.predictor = Conv2d(conv_dims, num_classes, kernel_size=1, stride=1, padding=0)
        c2_msra_fill(self.predictor)

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {
            "input_shape": {k: v for k, v in input_shape.items() if k in cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES},
            "ignore_value": cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE,
            "num_classes": cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
            "conv_dims": cfg.MODEL.SEM_SEG_HEAD.CONVS_DIM,
            "common_stride": cfg.MODEL.SEM_SEG_HEAD.COMMON_STRIDE,
            "norm": cfg.MODEL.SEM_SEG_HEAD.NORM,
            "loss_weight": cfg.MODEL.SEM_SEG_HEAD.LOSS_WEIGHT,
        }

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

    def forward(self, features, targets_u8=None):
        """targets_u8: [B, H, W] uint8 (ignore_value where unlabeled).  Training -> (None, {"loss_sem_seg"});
        inference -> ([B, num_classes, H, W] fp32 logits, {})."""
        x = self.layers(features)
        if self.training:
            loss = F.sem_seg_loss(x, targets_u8, self.num_classes, self.ignore_value)
            return None, {"loss_sem_seg": loss * self.loss_weight}
        # x4 bilinear (align_corners=False) of the fp32 logits and their argmax in one kernel; the argmax rides along on the
        # result tensor for PanopticFPN.inference (panoptic_fpn.py:173), which would otherwise read the logits back
        logits, argmax = F.sem_seg_upsample(x, self.num_classes, self.common_stride)
        logits.u2_argmax = argmax
        return logits, {}


def build_sem_seg_head(cfg, input_shape):
    return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, input_shape)
