"""Instance predictions in COCO result format and the cluster -> category mapping step of the U2Seg instance evaluation
(detectron2/evaluation/coco_evaluation.py:40-213, 227-335, 590-640).

mode "hungarian_matching": confident detections (score >= 0.6) vote for the category of every ground-truth box they
overlap with IoU > 0.7; the majority mapping of the 300 evaluated clusters is written to
./hungarian_matching/instance_mapping.json (the reference then exits).  mode "eval": the mapping is applied, detections of
unmapped clusters are dropped, category ids go back to dataset ids, the results are written in COCO format and scored by
evaluation/cocoeval.py (box AP / AR without pycocotools; its matching and accumulation are pinned to the reference's C++
evaluation core)."""
import itertools
import json
import os

import numpy as np
import torch

from ..data import rle
from ..data.catalog import MetadataCatalog
from ..data.detection_utils import BoxMode
from . import hungarian
from .evaluator import DatasetEvaluator, gather_to_rank0

SCORE_THRESH = 0.6
IOU_THRESH = 0.7
NUM_EVAL_CLUSTERS = 300
NUM_GT_CLASSES = 80


def instances_to_coco_json(instances, img_id):
    """[{"image_id", "category_id", "bbox": [x, y, w, h], "score", "segmentation": compressed RLE}] for one image."""
    n = len(instances)
    if n == 0:
        return []
    boxes = BoxMode.convert(instances.pred_boxes.tensor.numpy(), BoxMode.XYXY_ABS, BoxMode.XYWH_ABS).tolist()
    scores = instances.scores.tolist()
    classes = instances.pred_classes.tolist()
    rles = None
    if instances.has("pred_masks"):
        rles = [rle.encode(np.asarray(m, dtype=np.uint8)) for m in instances.pred_masks.numpy()]
    results = []
    for k in range(n):
        r = {"image_id": img_id, "category_id": classes[k], "bbox": boxes[k], "score": scores[k]}
        if rles is not None:
            r["segmentation"] = rles[k]
        results.append(r)
    return results


class COCOEvaluator(DatasetEvaluator):
    def __init__(self, dataset_name, output_dir=None, *, mode="hungarian_matching",
                 mapping_path="./hungarian_matching/instance_mapping.json"):
        self._metadata = MetadataCatalog.get(dataset_name)
        self._output_dir = output_dir
        self.mode = mode
        self.hungarain_matching_save_path = mapping_path
        data = json.load(open(self._metadata.json_file))
        self._img_to_anns = {}
        for ann in data.get("annotations", []):
            self._img_to_anns.setdefault(ann["image_id"], []).append(ann)
        self._cpu = torch.device("cpu")
        self.reset()

    def reset(self):
        self._predictions = []

    def process(self, inputs, outputs):
        for inp, out in zip(inputs, outputs):
            if "instances" in out:
                inst = out["instances"].to(self._cpu)
                self._predictions.append({"image_id": inp["image_id"],
                                          "instances": instances_to_coco_json(inst, inp["image_id"])})

    def cluster_mapping(self, coco_results, num_clusters=NUM_EVAL_CLUSTERS):
        """do_hangarain_mapping (:227-271) without the file write."""
        gt_id = dict(self._metadata.thing_dataset_id_to_contiguous_id)
        preds, targets = [], []
        for r in coco_results:
            if r["score"] < SCORE_THRESH:
                continue
            anns = self._img_to_anns.get(r["image_id"], [])
            if not anns:
                continue
            ious = hungarian.box_iou_xywh(r["bbox"], [a["bbox"] for a in anns])
            for a, iou in zip(anns, ious.tolist()):
                if iou > IOU_THRESH:
                    targets.append(gt_id[a["category_id"]])
                    preds.append(r["category_id"])
        return hungarian.majority_vote_mapping(preds, targets, range(num_clusters), NUM_GT_CLASSES)

    def evaluate(self):
        parts = gather_to_rank0(self._predictions)
        if parts is None:
            return {}  # only the main process evaluates (:189-197)
        predictions = list(itertools.chain(*parts))
        coco_results = list(itertools.chain(*[p["instances"] for p in predictions]))
        if self.mode == "hungarian_matching":
            mapping = self.cluster_mapping(coco_results)
            hungarian.save_mapping(mapping, self.hungarain_matching_save_path)
            return {"instance_mapping": mapping}
        mapping = hungarian.load_mapping(self.hungarain_matching_save_path)
        to_dataset = {v: k for k, v in self._metadata.thing_dataset_id_to_contiguous_id.items()}
        remapped = []
        for r in coco_results:
            c = mapping.get(r["category_id"], -1)
            if c == -1:
                continue
            r = dict(r)
            r["category_id"] = to_dataset[c]
            remapped.append(r)
        if self._output_dir:
            os.makedirs(self._output_dir, exist_ok=True)
            with open(os.path.join(self._output_dir, "coco_instances_results.json"), "w") as f:
                json.dump(remapped, f)
        results = {"num_results": len(remapped), "num_dropped": len(coco_results) - len(remapped)}
        results.update(self._box_metrics(remapped))
        return {"bbox": results}

    def _box_metrics(self, coco_results):
        """AP, AP50, AP75, APs, APm, APl (x 100, NaN where undefined) and the per-category APs, as _derive_coco_results
        reports them (:473-540; the reference skips the "segm" task, :346-347), from evaluation/cocoeval.py."""
        from . import cocoeval

        names = ("AP", "AP50", "AP75", "APs", "APm", "APl")
        if not coco_results:
            return {n: float("nan") for n in names}  # "No predictions from the model!"
        dataset = json.load(open(self._metadata.json_file))
        out = cocoeval.evaluate_bbox(dataset, coco_results)
        res = {n: (out["stats"][n] * 100 if out["stats"][n] >= 0 else float("nan")) for n in names}
        cats = sorted(dataset["categories"], key=lambda c: c["id"])
        if len(cats) > 1:
            for k, cat in enumerate(cats):
                p = out["precision"][:, :, k, 0, -1]
                p = p[p > -1]
                res["AP-" + str(cat.get("name", cat["id"]))] = float(p.mean() * 100) if p.size else float("nan")
        return res
