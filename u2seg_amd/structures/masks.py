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


def crop_and_resize_batch(bitmasks_list, boxes_list, mask_size):
    """BitMasks.crop_and_resize of several images (structures/masks.py:191-218) in ONE kernel launch: the bitmaps of
    each image stay where they are (a table of base pointers goes to the kernel), the lazy row selections become the
    per-ROI bitmap index.  Returns bool [sum_i len(boxes_i), mask_size, mask_size]."""
    import ctypes

    from .. import _hip
    from ..modeling.batched import image_index

    n = len(bitmasks_list)
    sizes = [len(b) for b in boxes_list]
    total = sum(sizes)
    dev = bitmasks_list[0].device
    out = torch.empty((total, mask_size, mask_size), dtype=torch.uint8, device=dev)
    if total == 0:
        return out.to(torch.bool)
    bases, rows = [], []
    for bm, k in zip(bitmasks_list, sizes):
        assert len(bm) == k, "{} != {}".format(len(bm), k)
        bases.append(bm._base.contiguous())
        rows.append(torch.arange(k, device=dev) if bm._index is None else bm._index)
    boxes = torch.cat([b.float() for b in boxes_list], dim=0)
    rois = torch.cat([torch.cat(rows).to(torch.float32)[:, None], boxes], dim=1).contiguous()
    roi_image = image_index(sizes, dev).to(torch.int32)
    ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bases])
    hs = (ctypes.c_int * n)(*[b.shape[1] for b in bases])
    ws = (ctypes.c_int * n)(*[b.shape[2] for b in bases])
    _hip.call("u2_mask_crop_batch", ptrs, hs, ws, n, rois, roi_image, out, total, mask_size)
    return out.to(torch.bool)
