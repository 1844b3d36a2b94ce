"""Dataset dict -> model input with the contract of detectron2/data/dataset_mapper.py:20-191.

Stages: load (image in INPUT.FORMAT, optional label map, size check) -> augment (the policy draws its parameters from
numpy's global RNG and rewrites image and label map) -> tensors (CHW uint8 image, int64 label map) -> in training, the
non-crowd annotations pushed through the same transforms and packed into Instances (empty ones dropped).  The input dict
is never modified; keys that are not consumed ("file_name", "height", "width", "image_id", ...) pass through."""
import copy

import numpy as np
import torch

from . import detection_utils as utils
from . import transforms as T


class DatasetMapper:
    def __init__(self, cfg=None, is_train=True, *, augmentations=None, image_format=None, use_instance_mask=None,
                 instance_mask_format=None):
        """DatasetMapper(cfg, is_train) like the reference's config constructor, or explicit keyword arguments."""
        if cfg is not None:
            unsupported = cfg.INPUT.CROP.ENABLED or cfg.MODEL.KEYPOINT_ON or cfg.MODEL.LOAD_PROPOSALS
            assert not unsupported, "crop / keypoint / precomputed-proposal inputs are not part of the U2Seg configs"
            if augmentations is None:
                augmentations = utils.build_augmentation(cfg, is_train)
            image_format = image_format or cfg.INPUT.FORMAT
            use_instance_mask = cfg.MODEL.MASK_ON if use_instance_mask is None else use_instance_mask
            instance_mask_format = instance_mask_format or cfg.INPUT.MASK_FORMAT
        self.is_train = is_train
        self.augmentations = T.AugmentationList(augmentations)
        self.image_format = image_format
        self.use_instance_mask = bool(use_instance_mask)
        self.instance_mask_format = instance_mask_format or "polygon"

    # ---- stages ---------------------------------------------------------------------------------
    def _load(self, record):
        image = utils.read_image(record["file_name"], format=self.image_format)
        utils.check_image_size(record, image)  # also fills in "width" / "height" when the dict lacks them
        label_file = record.pop("sem_seg_file_name", None)
        labels = None if label_file is None else utils.read_image(label_file, "L").squeeze(2)
        return image, labels

    def _augment(self, image, labels):
        state = T.AugInput(image, sem_seg=labels)
        transforms = self.augmentations(state)
        return state.image, state.sem_seg, transforms

    def _instances(self, annotations, transforms, image_shape):
        kept = []
        for anno in annotations:
            if anno.get("iscrowd", 0) != 0:
                continue  # crowd regions do not train the detector
            anno.pop("keypoints", None)
            if not self.use_instance_mask:
                anno.pop("segmentation", None)
            kept.append(utils.transform_instance_annotations(anno, transforms, image_shape))
        instances = utils.annotations_to_instances(kept, image_shape, mask_format=self.instance_mask_format)
        return utils.filter_empty_instances(instances)

    # ---- the mapper -----------------------------------------------------------------------------
    def __call__(self, dataset_dict):
        record = copy.deepcopy(dataset_dict)
        image, labels, transforms = self._augment(*self._load(record))
        image_shape = image.shape[:2]
        record["image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        if labels is not None:
            record["sem_seg"] = torch.as_tensor(labels.astype("long"))
        annotations = record.pop("annotations", None)
        if self.is_train and annotations is not None:
            record["instances"] = self._instances(annotations, transforms, image_shape)
        return record
