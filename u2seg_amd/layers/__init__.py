from .modules import (BatchNorm2d, Conv2d, ConvTranspose2d, FrozenBatchNorm2d, GroupNorm, Linear, ShapeSpec,
                      c2_msra_fill, c2_xavier_fill, get_norm, to_nhwc, to_nchw)

__all__ = ["BatchNorm2d", "Conv2d", "ConvTranspose2d", "FrozenBatchNorm2d", "GroupNorm", "Linear", "ShapeSpec",
           "c2_msra_fill", "c2_xavier_fill", "get_norm", "to_nhwc", "to_nchw"]
