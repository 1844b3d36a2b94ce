from .build import (DevicePrefetcher, InferenceSampler, TrainingSampler, build_detection_test_loader,
                    build_detection_train_loader, get_detection_dataset_dicts)
from .catalog import DatasetCatalog, MetadataCatalog
from .dataset_mapper import DatasetMapper
from .datasets import register_all_coco, register_coco_instances, register_coco_panoptic_separated
from .synthetic import make_synthetic_batch, synthetic_sample

__all__ = ["DatasetCatalog", "DatasetMapper", "DevicePrefetcher", "InferenceSampler", "MetadataCatalog", "TrainingSampler",
           "build_detection_test_loader", "build_detection_train_loader", "get_detection_dataset_dicts",
           "make_synthetic_batch", "register_all_coco", "register_coco_instances", "register_coco_panoptic_separated",
           "synthetic_sample"]
