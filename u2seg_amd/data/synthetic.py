"""Seeded synthetic COCO-panoptic-shaped samples in the DatasetMapper output format
(detectron2/data/dataset_mapper.py:144-191): {"image": uint8 3xHxW, "instances": Instances(gt_boxes, gt_classes,
gt_masks), "sem_seg": int64 HxW, "height", "width"}.  Recipe from SURVEY.md section 8(d)."""
import torch

from ..structures import BitMasks, Boxes, Instances


def synthetic_sample(index, height=800, width=1333, num_thing_classes=800, num_stuff_classes=28, min_inst=3,
                     max_inst=12, ignore_frac=0.05, device="cpu"):
    g = torch.Generator().manual_seed(1000 + index)
    image = torch.randint(0, 256, (3, height, width), generator=g, dtype=torch.uint8)
    # piecewise-constant 64x64 block labels, ~5% ignore
    bh, bw = (height + 63) // 64, (width + 63) // 64
    blocks = torch.randint(0, num_stuff_classes, (bh, bw), generator=g)
    sem = blocks.repeat_interleave(64, 0).repeat_interleave(64, 1)[:height, :width].contiguous()
    ign = torch.rand((height, width), generator=g) < ignore_frac
    sem[ign] = 255
    n = int(torch.randint(min_inst, max_inst + 1, (1,), generator=g))
    bw_ = 32 + torch.rand(n, generator=g) * (0.5 * width - 32)
    bh_ = 32 + torch.rand(n, generator=g) * (0.5 * height - 32)
    x0 = torch.rand(n, generator=g) * (width - bw_)
    y0 = torch.rand(n, generator=g) * (height - bh_)
    boxes = torch.stack([x0, y0, x0 + bw_, y0 + bh_], dim=1).float()
    classes = torch.randint(0, num_thing_classes, (n,), generator=g)
    ys = torch.arange(height, dtype=torch.float32)[None, :, None] + 0.5
    xs = torch.arange(width, dtype=torch.float32)[None, None, :] + 0.5
    cx, cy = ((boxes[:, 0] + boxes[:, 2]) / 2)[:, None, None], ((boxes[:, 1] + boxes[:, 3]) / 2)[:, None, None]
    rx, ry = (bw_ / 2)[:, None, None], (bh_ / 2)[:, None, None]
    masks = (((xs - cx) / rx) ** 2 + ((ys - cy) / ry) ** 2) <= 1.0
    inst = Instances((height, width))
    inst.gt_boxes = Boxes(boxes)
    inst.gt_classes = classes
    inst.gt_masks = BitMasks(masks)
    sample = {"image": image, "instances": inst, "sem_seg": sem, "height": height, "width": width}
    if device != "cpu":
        sample = {k: (v.to(device) if hasattr(v, "to") else v) for k, v in sample.items()}
    return sample


def make_synthetic_batch(batch_size, start_index=0, device="cpu", **kwargs):
    return [synthetic_sample(start_index + i, device=device, **kwargs) for i in range(batch_size)]
