from .boxes import Boxes, pairwise_iou
from .image_list import ImageList
from .instances import Instances
from .masks import BitMasks

__all__ = ["Boxes", "pairwise_iou", "ImageList", "Instances", "BitMasks"]
