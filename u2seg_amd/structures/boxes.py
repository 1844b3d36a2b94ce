"""Nx4 XYXY box container exposing the part of detectron2.structures.Boxes the hot path and the data mappers call
(reference API: detectron2/structures/boxes.py:130-358 - names and semantics only; written for this package: every geometric
method works on the (N, 2, 2) corner view of the tensor instead of on its four columns)."""
import torch


def _as_box_tensor(value):
    """float32 [N, 4]; an empty input of any shape becomes [0, 4]."""
    t = value if isinstance(value, torch.Tensor) else torch.as_tensor(value, dtype=torch.float32)
    t = t.to(torch.float32)
    if t.numel() == 0:
        t = t.reshape(0, 4)
    if t.dim() != 2 or t.shape[1] != 4:
        raise AssertionError("boxes must be [N, 4], got %s" % (tuple(t.shape),))
    return t


class Boxes:
    def __init__(self, tensor):
        self.tensor = _as_box_tensor(tensor)

    # ---- views -------------------------------------------------------------------------------------------------
    def _corners(self):
        """[N, 2 (min / max corner), 2 (x, y)] view of the storage."""
        return self.tensor.unflatten(1, (2, 2))

    def _extent(self):
        """[N, 2] widths and heights."""
        c = self._corners()
        return c[:, 1] - c[:, 0]

    # ---- detectron2 surface ------------------------------------------------------------------------------------
    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, device):
        return Boxes(self.tensor.to(device=device))

    @property
    def device(self):
        return self.tensor.device

    def area(self):
        return self._extent().prod(dim=1)

    def clip(self, box_size):
        """Clamp in place to the image rectangle [0, w] x [0, h]."""
        if not bool(torch.isfinite(self.tensor).all()):
            raise AssertionError("cannot clip boxes that contain Inf or NaN")
        h, w = box_size
        # Python-scalar bounds per coordinate column: no host -> device upload (a pageable copy would block the host)
        out = self.tensor.clamp(min=0)   # a fresh tensor: the strided column slices below are views of it
        out[:, 0::2].clamp_(max=w)
        out[:, 1::2].clamp_(max=h)
        self.tensor = out

    def nonempty(self, threshold=0.0):
        return (self._extent() > threshold).all(dim=1)

    def get_centers(self):
        return self._corners().mean(dim=1)

    def scale(self, scale_x, scale_y):
        # strided column slices (structures/boxes.py:212-217 indexes the same way): works on non-contiguous box tensors too
        self.tensor[:, 0::2].mul_(scale_x)
        self.tensor[:, 1::2].mul_(scale_y)

    def __getitem__(self, item):
        picked = self.tensor[item]
        if isinstance(item, int):
            picked = picked.unsqueeze(0)
        if picked.dim() != 2:
            raise AssertionError("index %r selects a %d-d tensor from Boxes, expected rows of 4" % (item, picked.dim()))
        return Boxes(picked)

    def __len__(self):
        return int(self.tensor.shape[0])

    def __iter__(self):
        return iter(self.tensor)

    def __repr__(self):
        return "Boxes(%s)" % (self.tensor,)

    @classmethod
    def cat(cls, boxes_list):
        if not isinstance(boxes_list, (list, tuple)):
            raise AssertionError("Boxes.cat takes a list or tuple of Boxes")
        rows = [b.tensor for b in boxes_list]
        return cls(torch.cat(rows, dim=0) if rows else torch.empty((0, 4)))


def pairwise_iou(boxes1, boxes2):
    """IoU matrix [N, M] (structures/boxes.py:312-358); plain-torch utility for callers outside the hot path."""
    a, b = boxes1.tensor, boxes2.tensor
    area1, area2 = boxes1.area(), boxes2.area()
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (area1[:, None] + area2 - inter), torch.zeros(1, dtype=inter.dtype, device=inter.device))
