"""Bitmap instance masks (detectron2/structures/masks.py:88-218)."""
import torch


class BitMasks:
    def __init__(self, tensor):
        if isinstance(tensor, torch.Tensor):
            tensor = tensor.to(torch.bool)
        else:
            tensor = torch.as_tensor(tensor, dtype=torch.bool, device=torch.device("cpu"))
        assert tensor.dim() == 3, tensor.size()
        self.image_size = tensor.shape[1:]
        self.tensor = tensor

    def to(self, *args, **kwargs):
        return BitMasks(self.tensor.to(*args, **kwargs))

    @property
    def device(self):
        return self.tensor.device

    def __getitem__(self, item):
        if isinstance(item, int):
            return BitMasks(self.tensor[item].unsqueeze(0))
        m = self.tensor[item]
        assert m.dim() == 3
        return BitMasks(m)

    def __len__(self):
        return self.tensor.shape[0]

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
        device = self.tensor.device
        idx = torch.arange(len(boxes), device=device).to(dtype=boxes.dtype)[:, None]
        rois = torch.cat([idx, boxes], dim=1)
        return F.mask_crop(self.tensor.to(torch.uint8), rois.float(), mask_size).to(torch.bool)
