"""Image reading and annotation handling of the input pipeline (detectron2/data/detection_utils.py:60-211, 257-331,
382-455, 486-520, 629-655; structures/boxes.py:15-131 for BoxMode)."""
import enum

import numpy as np
import torch
from PIL import Image

from ..structures import BitMasks, Boxes, Instances
from . import rle
from . import transforms as T

_EXIF_ORIENT = 274  # exif 'Orientation' tag


class SizeMismatchError(ValueError):
    """The loaded image has a different width / height than the annotation says."""


class BoxMode(enum.IntEnum):
    XYXY_ABS = 0
    XYWH_ABS = 1

    @staticmethod
    def convert(box, from_mode, to_mode):
        """Returns the type it was given.  A list / tuple goes through torch.tensor(box) like the reference
        (structures/boxes.py:62-131), i.e. json floats are added in fp32 and json ints stay ints - gt boxes are
        bit-identical only if this detail is kept; ndarrays keep their dtype."""
        if from_mode == to_mode:
            return box
        single = isinstance(box, (list, tuple))
        if single:
            assert len(box) == 4, "BoxMode.convert takes either a k-tuple/list or an Nxk array/tensor, where k == 4"
            arr = torch.tensor(box)[None, :]
        elif isinstance(box, np.ndarray):
            arr = torch.from_numpy(np.asarray(box)).clone()
        else:
            arr = box.clone()
        if from_mode == BoxMode.XYWH_ABS and to_mode == BoxMode.XYXY_ABS:
            arr[:, 2] += arr[:, 0]
            arr[:, 3] += arr[:, 1]
        elif from_mode == BoxMode.XYXY_ABS and to_mode == BoxMode.XYWH_ABS:
            arr[:, 2] -= arr[:, 0]
            arr[:, 3] -= arr[:, 1]
        else:
            raise NotImplementedError("Conversion from BoxMode {} to {} is not supported".format(from_mode, to_mode))
        if single:
            return type(box)(arr.flatten().tolist())
        return arr.numpy() if isinstance(box, np.ndarray) else arr


def _apply_exif_orientation(image):
    try:
        exif = image.getexif()
    except Exception:
        return image
    method = {2: Image.FLIP_LEFT_RIGHT, 3: Image.ROTATE_180, 4: Image.FLIP_TOP_BOTTOM, 5: Image.TRANSPOSE,
              6: Image.ROTATE_270, 7: Image.TRANSVERSE, 8: Image.ROTATE_90}.get(exif.get(_EXIF_ORIENT) if exif else None)
    return image.transpose(method) if method is not None else image


def convert_PIL_to_numpy(image, format):
    """PIL image -> HWC (or HW1 for "L") uint8 array in `format`; "BGR" is RGB with the channel axis reversed."""
    mode = "RGB" if format == "BGR" else format
    pixels = np.asarray(image.convert(mode) if mode is not None else image)
    if format == "L":
        return pixels[..., None]
    return pixels[..., ::-1] if format == "BGR" else pixels


def read_image(file_name, format=None):
    """HWC uint8 array in `format` ("RGB", "BGR", "L", any PIL mode), exif orientation applied."""
    assert format != "YUV-BT.601", "not used by the U2Seg configs (INPUT.FORMAT is RGB)"
    with open(file_name, "rb") as f:
        image = Image.open(f)
        image = _apply_exif_orientation(image)
        return convert_PIL_to_numpy(image, format)


def check_image_size(dataset_dict, image):
    """The annotation's idea of the image size must match the file; a dict without a size receives the file's."""
    h, w = image.shape[:2]
    if "width" in dataset_dict or "height" in dataset_dict:
        stated = (dataset_dict["width"], dataset_dict["height"])
        if stated != (w, h):
            where = " for image " + dataset_dict["file_name"] if "file_name" in dataset_dict else ""
            raise SizeMismatchError("the annotation%s states width x height = %s, the file has %s; check the json"
                                    % (where, stated, (w, h)))
    dataset_dict.setdefault("width", w)
    dataset_dict.setdefault("height", h)


def transform_instance_annotations(annotation, transforms, image_size):
    """Box through apply_box + clip to the new image, RLE / bitmap masks through apply_segmentation, polygons through
    apply_polygons (detection_utils.py:270-331); bbox_mode becomes XYXY_ABS.  Modifies and returns `annotation`."""
    if isinstance(transforms, (tuple, list)):
        transforms = T.TransformList(transforms)
    corners = BoxMode.convert(annotation["bbox"], annotation["bbox_mode"], BoxMode.XYXY_ABS)
    moved = transforms.apply_box(np.array([corners]))[0]
    h, w = image_size
    annotation["bbox"], annotation["bbox_mode"] = np.clip(moved, 0, [w, h, w, h]), BoxMode.XYXY_ABS
    if "segmentation" in annotation:
        segm = annotation["segmentation"]
        if isinstance(segm, list):
            polygons = [np.asarray(p, dtype=np.float64).reshape(-1, 2) for p in segm]
            annotation["segmentation"] = [p.reshape(-1) for p in transforms.apply_polygons(polygons)]
        elif isinstance(segm, dict):
            # row-major before resizing: PIL reads a column-major view (what the RLE scan order yields) twice as slowly
            mask = transforms.apply_segmentation(np.ascontiguousarray(rle.decode(segm)))
            assert tuple(mask.shape[:2]) == image_size
            annotation["segmentation"] = mask
        else:
            raise ValueError("Cannot transform segmentation of type '{}'!Supported types are: polygons as list[list[float] "
                             "or ndarray], COCO-style RLE as a dict.".format(type(segm)))
    return annotation


def annotations_to_instances(annos, image_size, mask_format="polygon"):
    """detection_utils.py:382-455 for the bitmask format the U2Seg configs use."""
    boxes = (np.stack([BoxMode.convert(obj["bbox"], obj["bbox_mode"], BoxMode.XYXY_ABS) for obj in annos])
             if len(annos) else np.zeros((0, 4)))
    target = Instances(image_size)
    target.gt_boxes = Boxes(torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 4))
    target.gt_classes = torch.tensor([int(obj["category_id"]) for obj in annos], dtype=torch.int64)
    if len(annos) and "segmentation" in annos[0]:
        if mask_format != "bitmask":
            raise NotImplementedError("INPUT.MASK_FORMAT '%s': the U2Seg configs use 'bitmask' (RLE pseudo-labels); polygon "
                                      "rasterisation needs pycocotools" % mask_format)
        out = torch.empty((len(annos),) + tuple(image_size), dtype=torch.bool)
        view = out.numpy()  # written in place: one pass per mask, whatever its strides (flipped views, Fortran order)
        for i, obj in enumerate(annos):
            segm = obj["segmentation"]
            if isinstance(segm, dict):
                segm = rle.decode(segm)
            elif not isinstance(segm, np.ndarray):
                raise ValueError("Cannot convert segmentation of type '{}' to BitMasks!".format(type(segm)))
            assert segm.ndim == 2, "Expect segmentation of 2 dimensions, got {}.".format(segm.ndim)
            np.not_equal(segm, 0, out=view[i])
        target.gt_masks = BitMasks(out)
    return target


def filter_empty_instances(instances, by_box=True, by_mask=True, box_threshold=1e-5):
    """Drops instances whose box is thinner than box_threshold or whose mask has no pixel (detection_utils.py:486-520)."""
    assert by_box or by_mask
    keep = torch.ones(len(instances), dtype=torch.bool)
    if by_box:
        keep &= instances.gt_boxes.nonempty(threshold=box_threshold)
    if by_mask and instances.has("gt_masks"):
        keep &= instances.gt_masks.nonempty()
    return instances[keep]


def build_augmentation(cfg, is_train):
    """detection_utils.py:629-655: ResizeShortestEdge, followed in training by RandomFlip unless INPUT.RANDOM_FLIP is "none"."""
    inp = cfg.INPUT
    if is_train:
        resize = T.ResizeShortestEdge(inp.MIN_SIZE_TRAIN, inp.MAX_SIZE_TRAIN, inp.MIN_SIZE_TRAIN_SAMPLING)
    else:
        resize = T.ResizeShortestEdge(inp.MIN_SIZE_TEST, inp.MAX_SIZE_TEST, "choice")
    policy = [resize]
    if is_train and inp.RANDOM_FLIP != "none":
        policy.append(T.RandomFlip(horizontal=inp.RANDOM_FLIP == "horizontal", vertical=inp.RANDOM_FLIP == "vertical"))
    return policy
