"""Bitmap instance masks (detectron2/structures/masks.py:88-218).

Indexing with an integer tensor is lazy: the result shares the full-resolution bitmaps of its parent and only keeps the
row selection.  label_and_sample_proposals (roi_heads.py:246-333) indexes gt_masks once per sampled proposal
(512 x 800 x 1333 bytes per image if materialised); crop_and_resize hands the selection to the mask_crop kernel as the
per-ROI mask index instead, so the gathered copy is never formed unless `.tensor` is read."""
import torch


class BitMasks:
    def __init__(self, tensor, index=None):
        if isinstance(tensor, torch.Tensor):
            if tensor.dtype != torch.bool:
                tensor = tensor.to(torch.bool)
        else:
            tensor = torch.as_tensor(tensor, dtype=torch.bool, device=torch.device("cpu"))
        assert tensor.dim() == 3, tensor.size()
        self.image_size = tensor.shape[1:]
        self._base = tensor
        self._index = index  # None (all rows of _base, in order) or int64 [n] rows of _base

    @property
    def tensor(self):
        if self._index is not None:
            self._base = self._base[self._index]
            self._index = None
        return self._base

    def to(self, *args, **kwargs):
        index = None if self._index is None else self._index.to(*args, **{k: v for k, v in kwargs.items() if k != "dtype"})
        return BitMasks(self._base.to(*args, **kwargs), index)

    @property
    def device(self):
        return self._base.device

    def __getitem__(self, item):
        if isinstance(item, int):
            return BitMasks(self.tensor[item].unsqueeze(0))
        if isinstance(item, torch.Tensor) and item.dim() == 1 and item.device == self._base.device:
            if item.dtype == torch.bool:
                assert item.numel() == len(self)
                rows = torch.nonzero(item, as_tuple=True)[0]
            elif item.dtype in (torch.int64, torch.int32):
                rows = item.long()
            else:
                rows = None
            if rows is not None:
                return BitMasks(self._base, rows if self._index is None else self._index[rows])
        m = self.tensor[item]
        assert m.dim() == 3
        return BitMasks(m)

    def __len__(self):
        return self._base.shape[0] if self._index is None else self._index.numel()

    def nonempty(self):
        return self.tensor.flatten(1).any(dim=1)

    @staticmethod
    def cat(bitmasks_list):
        assert len(bitmasks_list) > 0
        return BitMasks(torch.cat([bm.tensor for bm in bitmasks_list], dim=0))

    def crop_and_resize(self, boxes, mask_size):
        """ROIAlign(mask_size, 1.0, 0, aligned=True) of each mask inside its box, thresholded at 0.5
        (masks.py:191-218).  Runs the HIP mask_crop kernel; masks must live on the GPU."""
        from ..layers import functional as F

        assert len(boxes) == len(self), "{} != {}".format(len(boxes), len(self))
        device = self._base.device
        rows = torch.arange(len(boxes), device=device) if self._index is None else self._index
        rois = torch.cat([rows.to(dtype=torch.float32)[:, None], boxes.float()], dim=1)
        return F.mask_crop(self._base.contiguous().view(torch.uint8), rois, mask_size).to(torch.bool)
