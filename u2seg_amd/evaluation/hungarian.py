"""Cluster id -> ground-truth category mapping used by the U2Seg evaluators ("Hungarian matching" in the reference's
vocabulary; the procedure itself is a majority vote, evaluation/coco_evaluation.py:273-297 and
sem_seg_evaluation.py:147-162): every matched (predicted cluster, ground-truth category) pair is a vote, a cluster maps to
the category it was matched with most often and to -1 if it never matched."""
import json
import os

import numpy as np


def majority_vote_mapping(all_preds, all_targets, labels, num_classes):
    """{label: argmax_c #{k: preds[k] == label and targets[k] == c}} over `labels`, -1 without votes; ties go to the
    smaller category (np.argmax)."""
    all_preds = np.asarray(all_preds, dtype=np.int64)
    all_targets = np.asarray(all_targets, dtype=np.int64)
    mapping = {}
    for i in labels:
        votes = np.bincount(all_targets[all_preds == i], minlength=num_classes)
        mapping[i] = -1 if votes.sum() == 0 else int(np.argmax(votes))
    return mapping


def box_iou_xywh(box, boxes):
    """IoU of one [x, y, w, h] box with each of `boxes` (cocoapi maskApi.c bbIou with iscrowd = 0)."""
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    a = np.asarray(box, dtype=np.float64)
    w = np.minimum(a[0] + a[2], b[:, 0] + b[:, 2]) - np.maximum(a[0], b[:, 0])
    h = np.minimum(a[1] + a[3], b[:, 1] + b[:, 3]) - np.maximum(a[1], b[:, 1])
    inter = np.where((w <= 0) | (h <= 0), 0.0, w * h)
    union = a[2] * a[3] + b[:, 2] * b[:, 3] - inter
    return inter / union


def save_mapping(mapping, path):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(mapping, f, ensure_ascii=False)


def load_mapping(path):
    """json turns the integer keys into strings; give them back as ints."""
    return {int(k): int(v) for k, v in json.load(open(path)).items()}
