"""Geometric transforms and the two augmentations of the U2Seg input pipeline
(detectron2/data/transforms/{transform.py:94-160, augmentation.py:80-352, augmentation_impl.py:82-201};
the base-class behaviour of fvcore.transforms.transform that they rely on is restated here: apply_box maps the four corners
and takes their bounding box, a TransformList applies its members in order and drops no-ops).

A transform is deterministic; an augmentation draws its parameters from numpy's global RNG exactly like the reference
(np.random.choice for the short edge, np.random.uniform for the flip), so a seeded stream gives the same crops."""
import sys

import numpy as np
from PIL import Image

_BOX_CORNERS = np.array([0, 1, 2, 1, 0, 3, 2, 3])


class Transform:
    def apply_image(self, img):
        raise NotImplementedError

    def apply_coords(self, coords):
        raise NotImplementedError

    def apply_segmentation(self, segmentation):
        return self.apply_image(segmentation)

    def apply_box(self, box):
        corners = np.asarray(box, dtype=np.float64).reshape(-1, 4)[:, _BOX_CORNERS].reshape(-1, 2)
        corners = self.apply_coords(corners).reshape(-1, 4, 2)
        return np.concatenate([corners.min(axis=1), corners.max(axis=1)], axis=1)

    def apply_polygons(self, polygons):
        return [self.apply_coords(p) for p in polygons]


class NoOpTransform(Transform):
    def apply_image(self, img):
        return img

    def apply_coords(self, coords):
        return coords


class HFlipTransform(Transform):
    def __init__(self, width):
        self.width = width

    def apply_image(self, img):
        return np.flip(img, axis=1)

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords


class VFlipTransform(Transform):
    def __init__(self, height):
        self.height = height

    def apply_image(self, img):
        return np.flip(img, axis=0)

    def apply_coords(self, coords):
        coords[:, 1] = self.height - coords[:, 1]
        return coords


class ResizeTransform(Transform):
    """transform.py:94-160: PIL resize for uint8 images (bilinear by default, nearest for label maps)."""

    def __init__(self, h, w, new_h, new_w, interp=None):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w
        self.interp = Image.BILINEAR if interp is None else interp

    def apply_image(self, img, interp=None):
        assert img.shape[:2] == (self.h, self.w), (img.shape, self.h, self.w)
        assert img.dtype == np.uint8, "the U2Seg pipeline only resizes uint8 images and label maps"
        single = img.ndim > 2 and img.shape[2] == 1
        pil = Image.fromarray(img[:, :, 0], mode="L") if single else Image.fromarray(img)
        out = np.asarray(pil.resize((self.new_w, self.new_h), self.interp if interp is None else interp))
        return out[:, :, None] if single else out

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords

    def apply_segmentation(self, segmentation):
        return self.apply_image(segmentation, interp=Image.NEAREST)


class TransformList(Transform):
    def __init__(self, transforms):
        flat = []
        for t in transforms:
            flat.extend(t.transforms if isinstance(t, TransformList) else [t])
        self.transforms = [t for t in flat if not isinstance(t, NoOpTransform)]

    def _chain(self, name, x):
        for t in self.transforms:
            x = getattr(t, name)(x)
        return x

    def apply_image(self, img):
        return self._chain("apply_image", img)

    def apply_coords(self, coords):
        return self._chain("apply_coords", coords)

    def apply_segmentation(self, segmentation):
        return self._chain("apply_segmentation", segmentation)

    def apply_box(self, box):
        return self._chain("apply_box", box)

    def apply_polygons(self, polygons):
        return self._chain("apply_polygons", polygons)

    def __len__(self):
        return len(self.transforms)

    def __getitem__(self, i):
        return self.transforms[i]


class AugInput:
    """augmentation.py:278-352: the image (+ optional boxes / label map) an augmentation policy looks at and rewrites."""

    def __init__(self, image, *, boxes=None, sem_seg=None):
        assert isinstance(image, np.ndarray) and image.dtype in (np.uint8, np.float32), type(image)
        assert image.ndim in (2, 3), image.ndim
        self.image, self.boxes, self.sem_seg = image, boxes, sem_seg

    def transform(self, tfm):
        self.image = tfm.apply_image(self.image)
        if self.boxes is not None:
            self.boxes = tfm.apply_box(self.boxes)
        if self.sem_seg is not None:
            self.sem_seg = tfm.apply_segmentation(self.sem_seg)


class Augmentation:
    def get_transform(self, image):
        raise NotImplementedError

    def __call__(self, aug_input):
        tfm = self.get_transform(aug_input.image)
        aug_input.transform(tfm)
        return tfm


class ResizeShortestEdge(Augmentation):
    """augmentation_impl.py:134-201: scale the short edge to a sampled length, cap the long edge at max_size."""

    def __init__(self, short_edge_length, max_size=sys.maxsize, sample_style="range", interp=Image.BILINEAR):
        assert sample_style in ("range", "choice"), sample_style
        self.is_range = sample_style == "range"
        if isinstance(short_edge_length, int):
            short_edge_length = (short_edge_length, short_edge_length)
        if self.is_range:
            assert len(short_edge_length) == 2, short_edge_length
        self.short_edge_length, self.max_size, self.interp = short_edge_length, max_size, interp

    def get_transform(self, image):
        h, w = image.shape[:2]
        if self.is_range:
            size = np.random.randint(self.short_edge_length[0], self.short_edge_length[1] + 1)
        else:
            size = np.random.choice(self.short_edge_length)
        if size == 0:
            return NoOpTransform()
        newh, neww = self.get_output_shape(h, w, size, self.max_size)
        return ResizeTransform(h, w, newh, neww, self.interp)

    @staticmethod
    def get_output_shape(oldh, oldw, short_edge_length, max_size):
        size = short_edge_length * 1.0
        scale = size / min(oldh, oldw)
        newh, neww = (size, scale * oldw) if oldh < oldw else (scale * oldh, size)
        if max(newh, neww) > max_size:
            scale = max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return int(newh + 0.5), int(neww + 0.5)


class RandomFlip(Augmentation):
    """augmentation_impl.py:82-112."""

    def __init__(self, prob=0.5, *, horizontal=True, vertical=False):
        if horizontal == vertical:
            raise ValueError("RandomFlip flips along exactly one axis: got horizontal=%r, vertical=%r (chain two RandomFlip "
                             "augmentations to draw both)" % (horizontal, vertical))
        self.prob, self.horizontal, self.vertical = prob, horizontal, vertical

    def get_transform(self, image):
        h, w = image.shape[:2]
        if np.random.uniform(0, 1.0) < self.prob:
            return HFlipTransform(w) if self.horizontal else VFlipTransform(h)
        return NoOpTransform()


class AugmentationList(Augmentation):
    """augmentation.py:244-275: each member sees the input as its predecessors left it."""

    def __init__(self, augs):
        self.augs = list(augs)

    def __call__(self, aug_input):
        tfms = []
        for a in self.augs:
            if isinstance(a, Transform):
                aug_input.transform(a)
                tfms.append(a)
            else:
                tfms.append(a(aug_input))
        return TransformList(tfms)
