"""Semantic evaluation on COCO's 15 stuff supercategories after mapping the 27 unsupervised classes onto them
(detectron2/evaluation/sem_seg_evaluation.py:37-356).

mode "hungarian_matching": every (predicted class, ground-truth supercategory) pair of an image whose masks overlap with
IoU > 0.15 is a vote; evaluate() writes the majority mapping to ./hungarian_matching/semantic_mapping.json.
mode "eval": predictions are mapped (unmapped classes become the extra "ignore" label 16), a 17 x 17 confusion matrix is
accumulated with one bincount per image on the device the predictions live on, and mIoU / fwIoU / mACC / pACC plus the
per-class numbers come out of it.
Boundary IoU needs OpenCV, which this image does not have; the reference switches it off in that case too (:103-109)."""
from collections import OrderedDict

import numpy as np
import torch
from PIL import Image

from ..data.catalog import DatasetCatalog, MetadataCatalog
from . import hungarian
from .evaluator import DatasetEvaluator, gather_to_rank0

SUPERCATEGORIES = ("textile", "building", "raw-material", "furniture-stuff", "floor", "plant", "food-stuff", "ground",
                   "structural", "water", "wall", "window", "ceiling", "sky", "solid")
# COCO panoptic's 53 stuff categories in contiguous order (banner, blanket, bridge, ... rug-merged) -> index of their
# supercategory in SUPERCATEGORIES, 1-based (the tables at :171-190 of the reference, folded into one)
STUFF_TO_SUPERCATEGORY = (1, 1, 2, 3, 4, 1, 4, 5, 6, 7, 8, 2, 4, 4, 9, 1, 8, 8, 8, 10, 8, 2, 8, 10, 4, 8, 4, 2, 1, 11, 11, 11,
                          11, 10, 12, 12, 6, 9, 13, 14, 4, 4, 5, 8, 15, 6, 8, 3, 7, 2, 15, 11, 1)
IOU_THRESH = 0.15
NUM_CLUSTERS = 27


def load_image_into_numpy_array(filename, dtype=None):
    with open(filename, "rb") as f:
        return np.array(Image.open(f), dtype=dtype)


def to_supercategories(gt):
    """Label map with contiguous stuff ids 1..53 (0 = things, 255 = ignore) -> supercategory ids 1..15 (0 and 255 kept)."""
    lut = np.arange(256, dtype=gt.dtype)
    lut[1:54] = STUFF_TO_SUPERCATEGORY
    return lut[gt]


class SemSegEvaluator(DatasetEvaluator):
    def __init__(self, dataset_name, output_dir=None, *, mode="hungarian_matching",
                 mapping_path="./hungarian_matching/semantic_mapping.json", sem_seg_loading_fn=load_image_into_numpy_array):
        self._dataset_name, self._output_dir = dataset_name, output_dir
        self.input_file_to_gt_file = {r["file_name"]: r["sem_seg_file_name"] for r in DatasetCatalog.get(dataset_name)}
        meta = MetadataCatalog.get(dataset_name)
        self._ignore_label = meta.ignore_label
        self._class_names = ["things"] + list(SUPERCATEGORIES)
        self._num_classes = 16
        self.sem_seg_loading_fn = sem_seg_loading_fn
        self.mode = mode
        self.hungarain_matching_save_path = mapping_path
        self.pseudo_gt_cate, self.pred_det_cate = [], []
        self._lut = None
        self.reset()

    def reset(self):
        self._conf_matrix = torch.zeros((self._num_classes + 1, self._num_classes + 1), dtype=torch.int64)

    def _collect_votes(self, pred, gt):
        """One joint histogram of (predicted class, ground-truth supercategory) gives every pairwise intersection; with the
        marginals that is every pairwise IoU at once - the reference masks the image once per pair (:208-221).  Runs on the
        device the prediction lives on; only the handful of votes comes back."""
        n = self._num_classes + 1
        classes = max(int(pred.max()) + 1, NUM_CLUSTERS + 1)
        joint = torch.bincount(pred.reshape(-1) * n + gt.reshape(-1), minlength=classes * n).view(classes, n)
        area_p, area_g = joint.sum(dim=1, keepdim=True), joint.sum(dim=0, keepdim=True)
        union = (area_p + area_g - joint).double()
        iou = torch.where(union > 0, joint.double() / union.clamp(min=1), torch.zeros_like(union))
        vote = (iou > IOU_THRESH) & (area_p > 0) & (area_g > 0)
        vote[0, :] = False                     # predicted class 0 stands for "things"
        vote[:, 0] = vote[:, n - 1] = False    # so does ground truth 0; 16 is the ignore label
        for p, g in torch.nonzero(vote).tolist():
            self.pred_det_cate.append(p)
            self.pseudo_gt_cate.append(g)

    def process(self, inputs, outputs):
        n = self._num_classes + 1
        for inp, out in zip(inputs, outputs):
            pred = out["sem_seg"].argmax(dim=0)  # stays on the model's device
            gt_np = to_supercategories(self.sem_seg_loading_fn(self.input_file_to_gt_file[inp["file_name"]], dtype=int))
            gt_np[gt_np == self._ignore_label] = self._num_classes
            gt = torch.from_numpy(gt_np).to(pred.device)
            if self.mode == "hungarian_matching":
                self._collect_votes(pred, gt)
                continue
            if self._lut is None:
                # The reference rewrites the prediction in place, one cluster after the other in the order of the json
                # file (:247-252), so a pixel moved to label t is moved again when cluster t's turn comes.  The same
                # composition on a lookup table: lut[v] is where original label v currently stands.
                lut = np.arange(256)
                for cls, tgt in hungarian.load_mapping(self.hungarain_matching_save_path).items():
                    lut[lut == cls] = self._num_classes if tgt == -1 else tgt
                self._lut = torch.from_numpy(lut)
            lut = self._lut.to(pred.device)
            if self._conf_matrix.device != pred.device:
                self._conf_matrix = self._conf_matrix.to(pred.device)
            self._conf_matrix += torch.bincount(n * lut[pred].reshape(-1) + gt.reshape(-1), minlength=n * n).view(n, n)

    def cluster_mapping(self):
        mapping = hungarian.majority_vote_mapping(self.pred_det_cate, self.pseudo_gt_cate, range(1, NUM_CLUSTERS + 1), 15)
        mapping[0] = 0
        return mapping

    def evaluate(self):
        if self.mode == "hungarian_matching":
            votes = gather_to_rank0((self.pred_det_cate, self.pseudo_gt_cate))
            if votes is None:
                return None
            self.pred_det_cate = [p for part in votes for p in part[0]]
            self.pseudo_gt_cate = [g for part in votes for g in part[1]]
            mapping = self.cluster_mapping()
            hungarian.save_mapping(mapping, self.hungarain_matching_save_path)
            return OrderedDict({"sem_seg": None, "semantic_mapping": mapping})
        mats = gather_to_rank0(self._conf_matrix.cpu().numpy())
        if mats is None:
            return None
        cm = self._conf_matrix = sum(mats[1:], mats[0].copy())
        acc = np.full(self._num_classes, np.nan, dtype=float)
        iou = np.full(self._num_classes, np.nan, dtype=float)
        tp = cm.diagonal()[:-1].astype(float)
        pos_gt = np.sum(cm[:-1, :-1], axis=0).astype(float)
        class_weights = pos_gt / np.sum(pos_gt)
        pos_pred = np.sum(cm[:-1, :-1], axis=1).astype(float)
        acc_valid = pos_gt > 0
        acc[acc_valid] = tp[acc_valid] / pos_gt[acc_valid]
        union = pos_gt + pos_pred - tp
        iou_valid = np.logical_and(acc_valid, union > 0)
        iou[iou_valid] = tp[iou_valid] / union[iou_valid]
        res = {"mIoU": 100 * np.sum(iou[iou_valid]) / np.sum(iou_valid),
               "fwIoU": 100 * np.sum(iou[iou_valid] * class_weights[iou_valid])}
        for i, name in enumerate(self._class_names):
            res["IoU-" + name] = 100 * iou[i]
        res["mACC"] = 100 * np.sum(acc[acc_valid]) / np.sum(acc_valid)
        res["pACC"] = 100 * np.sum(tp) / np.sum(pos_gt)
        for i, name in enumerate(self._class_names):
            res["ACC-" + name] = 100 * acc[i]
        if self._output_dir:
            import os

            os.makedirs(self._output_dir, exist_ok=True)
            torch.save(res, os.path.join(self._output_dir, "sem_seg_evaluation.pth"))
        return OrderedDict({"sem_seg": res})
