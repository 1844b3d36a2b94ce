from .build import FlatSGD, WarmupMultiStepLR, build_lr_scheduler, build_optimizer

__all__ = ["FlatSGD", "WarmupMultiStepLR", "build_lr_scheduler", "build_optimizer"]
