from .coco_evaluation import COCOEvaluator, instances_to_coco_json
from .evaluator import DatasetEvaluator, DatasetEvaluators, inference_on_dataset
from .panoptic_evaluation import COCOPanopticEvaluator
from .sem_seg_evaluation import SemSegEvaluator



def build_evaluator(cfg, dataset_name, output_folder=None, eval_mode="eval"):
    """tools/train_net.py:42-81 of the reference for the evaluator types the U2Seg datasets carry: semantic, instance and
    panoptic evaluators for "coco_panoptic_seg"."""
    import os

    from ..data.catalog import MetadataCatalog

    if output_folder is None:
        output_folder = os.path.join(cfg.OUTPUT_DIR, "inference")
    kind = MetadataCatalog.get(dataset_name).evaluator_type
    evaluators = []
    if kind in ("sem_seg", "coco_panoptic_seg"):
        evaluators.append(SemSegEvaluator(dataset_name, output_dir=output_folder, mode=eval_mode))
    if kind in ("coco", "coco_panoptic_seg"):
        evaluators.append(COCOEvaluator(dataset_name, output_dir=output_folder, mode=eval_mode))
    if kind == "coco_panoptic_seg":
        evaluators.append(COCOPanopticEvaluator(dataset_name, output_folder))
    if not evaluators:
        raise NotImplementedError("no Evaluator for the dataset {} with the type {}".format(dataset_name, kind))
    return evaluators[0] if len(evaluators) == 1 else DatasetEvaluators(evaluators)


__all__ = ["build_evaluator", "COCOPanopticEvaluator", "COCOEvaluator", "DatasetEvaluator", "DatasetEvaluators", "SemSegEvaluator", "inference_on_dataset",
           "instances_to_coco_json"]
