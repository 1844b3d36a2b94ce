"""u2seg_amd: MI355X-native implementation of U2Seg's Panoptic-FPN hot path (HIP kernels behind the
Detectron2 registry / config API)."""
__version__ = "0.1.0"
