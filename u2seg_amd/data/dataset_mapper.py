"""Dataset dict -> model input (detectron2/data/dataset_mapper.py:20-191): read the image (and the label map), apply the
augmentations, turn the annotations into Instances."""
import copy

import numpy as np
import torch

from . import detection_utils as utils
from . import transforms as T


class DatasetMapper:
    def __init__(self, cfg=None, is_train=True, *, augmentations=None, image_format=None, use_instance_mask=None,
                 instance_mask_format=None):
        """Either from a config (DatasetMapper(cfg, is_train)) like the reference's @configurable constructor or from
        explicit keyword arguments."""
        if cfg is not None:
            assert not cfg.INPUT.CROP.ENABLED and not cfg.MODEL.KEYPOINT_ON and not cfg.MODEL.LOAD_PROPOSALS, \
                "crop / keypoint / precomputed-proposal inputs are not part of the U2Seg configs"
            augmentations = utils.build_augmentation(cfg, is_train) if augmentations is None else augmentations
            image_format = cfg.INPUT.FORMAT if image_format is None else image_format
            use_instance_mask = cfg.MODEL.MASK_ON if use_instance_mask is None else use_instance_mask
            instance_mask_format = cfg.INPUT.MASK_FORMAT if instance_mask_format is None else instance_mask_format
        self.is_train = is_train
        self.augmentations = T.AugmentationList(augmentations)
        self.image_format = image_format
        self.use_instance_mask = bool(use_instance_mask)
        self.instance_mask_format = instance_mask_format or "polygon"

    def _transform_annotations(self, dataset_dict, transforms, image_shape):
        for anno in dataset_dict["annotations"]:
            if not self.use_instance_mask:
                anno.pop("segmentation", None)
            anno.pop("keypoints", None)
        annos = [utils.transform_instance_annotations(obj, transforms, image_shape)
                 for obj in dataset_dict.pop("annotations") if obj.get("iscrowd", 0) == 0]
        instances = utils.annotations_to_instances(annos, image_shape, mask_format=self.instance_mask_format)
        dataset_dict["instances"] = utils.filter_empty_instances(instances)

    def __call__(self, dataset_dict):
        dataset_dict = copy.deepcopy(dataset_dict)
        image = utils.read_image(dataset_dict["file_name"], format=self.image_format)
        utils.check_image_size(dataset_dict, image)
        if "sem_seg_file_name" in dataset_dict:
            sem_seg_gt = utils.read_image(dataset_dict.pop("sem_seg_file_name"), "L").squeeze(2)
        else:
            sem_seg_gt = None
        aug_input = T.AugInput(image, sem_seg=sem_seg_gt)
        transforms = self.augmentations(aug_input)
        image, sem_seg_gt = aug_input.image, aug_input.sem_seg
        image_shape = image.shape[:2]
        dataset_dict["image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        if sem_seg_gt is not None:
            dataset_dict["sem_seg"] = torch.as_tensor(sem_seg_gt.astype("long"))
        if not self.is_train:
            dataset_dict.pop("annotations", None)
            dataset_dict.pop("sem_seg_file_name", None)
            return dataset_dict
        if "annotations" in dataset_dict:
            self._transform_annotations(dataset_dict, transforms, image_shape)
        return dataset_dict
