"""Batch of images padded to a common size (detectron2/structures/image_list.py:59-129)."""
import torch


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def padded_size(image_sizes, size_divisibility=0):
        """max (h, w) over the batch rounded up to the stride (image_list.py:88-101)."""
        mh = max(s[0] for s in image_sizes)
        mw = max(s[1] for s in image_sizes)
        if size_divisibility > 1:
            st = size_divisibility
            mh = (mh + (st - 1)) // st * st
            mw = (mw + (st - 1)) // st * st
        return mh, mw

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        assert len(tensors) > 0
        image_sizes = [(im.shape[-2], im.shape[-1]) for im in tensors]
        mh, mw = ImageList.padded_size(image_sizes, size_divisibility)
        batch_shape = [len(tensors)] + list(tensors[0].shape[:-2]) + [mh, mw]
        batched = tensors[0].new_full(batch_shape, pad_value)
        for i, img in enumerate(tensors):
            batched[i, ..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), image_sizes)
