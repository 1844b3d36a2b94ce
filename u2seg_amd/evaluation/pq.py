"""Panoptic quality (PQ = SQ x RQ) from COCO-panoptic-format predictions and ground truth.

PARITY UNPINNED: the reference delegates this to `panopticapi.evaluation.pq_compute` (panoptic_evaluation.py:160-168), a
third-party package (cocodataset/panopticapi, unpinned in the reference, absent from this image), and holds no test or
golden vector for it.  What follows restates the published procedure (Kirillov et al., "Panoptic Segmentation", CVPR 2019,
section 4; panopticapi/evaluation.py `pq_compute_single_core`) and is checked on hand-computable cases only; when
panopticapi is importable `COCOPanopticEvaluator` calls it instead.

Per image: segments are compared through the joint histogram of (ground-truth id, predicted id) over the pixels; a
predicted and a ground-truth segment of the same category match when their IoU exceeds 0.5 (at most one match each, which
the threshold guarantees), where pixels of the predicted segment that fall on unlabelled ground truth (id 0) do not count
towards the union; crowd ground truth never matches.  Unmatched ground truth (not crowd) is a false negative; an
unmatched prediction is a false positive unless more than half of it lies on void or on crowd regions of its own category.
Per category PQ = sum(IoU of matches) / (TP + FP / 2 + FN / 2); the reported numbers average the categories that occur."""
import numpy as np

VOID = 0
OFFSET = 256 * 256 * 256


class PQStat:
    def __init__(self):
        self.iou, self.tp, self.fp, self.fn = {}, {}, {}, {}

    def add(self, table, cat, value=1):
        table[cat] = table.get(cat, 0) + value

    def average(self, categories, isthing=None):
        pq = sq = rq = n = 0
        per_class = {}
        for cat_id, info in categories.items():
            if isthing is not None and bool(info["isthing"]) != isthing:
                continue
            tp, fp, fn = self.tp.get(cat_id, 0), self.fp.get(cat_id, 0), self.fn.get(cat_id, 0)
            if tp + fp + fn == 0:
                per_class[cat_id] = {"pq": 0.0, "sq": 0.0, "rq": 0.0}
                continue
            iou = self.iou.get(cat_id, 0.0)
            c = {"pq": iou / (tp + 0.5 * fp + 0.5 * fn), "sq": iou / tp if tp else 0.0, "rq": tp / (tp + 0.5 * fp + 0.5 * fn)}
            per_class[cat_id] = c
            n += 1
            pq, sq, rq = pq + c["pq"], sq + c["sq"], rq + c["rq"]
        if n == 0:
            return {"pq": 0.0, "sq": 0.0, "rq": 0.0, "n": 0}, per_class
        return {"pq": pq / n, "sq": sq / n, "rq": rq / n, "n": n}, per_class


def accumulate_image(stat, gt_ids, gt_segments, pred_ids, pred_segments, categories):
    """gt_ids / pred_ids: integer id maps [h, w] (0 = void); *_segments: COCO panoptic segments_info lists."""
    gt_ids, pred_ids = np.asarray(gt_ids, dtype=np.uint64), np.asarray(pred_ids, dtype=np.uint64)
    gt_seg = {s["id"]: s for s in gt_segments}
    pred_seg = {s["id"]: dict(s) for s in pred_segments}
    labels, counts = np.unique(pred_ids, return_counts=True)
    for label, cnt in zip(labels.tolist(), counts.tolist()):
        if label == VOID:
            continue
        if label not in pred_seg:
            raise KeyError("segment id %d is in the predicted png but not in segments_info" % label)
        if pred_seg[label]["category_id"] not in categories:
            raise KeyError("segment %d has unknown category %r" % (label, pred_seg[label]["category_id"]))
        pred_seg[label]["area"] = cnt
    missing = set(pred_seg) - set(labels.tolist())
    if missing:
        raise KeyError("segments_info lists ids that are not in the predicted png: %s" % sorted(missing))
    gt_area = dict(zip(*[x.tolist() for x in np.unique(gt_ids, return_counts=True)]))
    pairs, inter = np.unique(gt_ids * np.uint64(OFFSET) + pred_ids, return_counts=True)
    overlap = {(int(p) // OFFSET, int(p) % OFFSET): int(c) for p, c in zip(pairs, inter)}
    gt_matched, pred_matched = set(), set()
    for (g, p), area in overlap.items():
        if g not in gt_seg or p not in pred_seg:
            continue
        if gt_seg[g].get("iscrowd", 0) == 1 or gt_seg[g]["category_id"] != pred_seg[p]["category_id"]:
            continue
        # the published procedure takes the ground-truth area from the annotation json when it carries one
        union = pred_seg[p]["area"] + gt_seg[g].get("area", gt_area[g]) - area - overlap.get((VOID, p), 0)
        iou = area / union
        if iou > 0.5:
            cat = gt_seg[g]["category_id"]
            stat.add(stat.tp, cat)
            stat.add(stat.iou, cat, iou)
            gt_matched.add(g)
            pred_matched.add(p)
    crowd_of = {}
    for g, info in gt_seg.items():
        if g in gt_matched:
            continue
        if info.get("iscrowd", 0) == 1:
            crowd_of[info["category_id"]] = g
            continue
        stat.add(stat.fn, info["category_id"])
    for p, info in pred_seg.items():
        if p in pred_matched:
            continue
        excused = overlap.get((VOID, p), 0)
        if info["category_id"] in crowd_of:
            excused += overlap.get((crowd_of[info["category_id"]], p), 0)
        if excused / info["area"] > 0.5:
            continue
        stat.add(stat.fp, info["category_id"])


def pq_compute_arrays(samples, categories):
    """samples: iterable of (gt_ids, gt_segments, pred_ids, pred_segments); categories: {id: {"isthing": 0 | 1}}.
    Returns {"All" | "Things" | "Stuff": {"pq", "sq", "rq", "n"}, "per_class": {...}} like pq_compute."""
    stat = PQStat()
    for gt_ids, gt_segments, pred_ids, pred_segments in samples:
        accumulate_image(stat, gt_ids, gt_segments, pred_ids, pred_segments, categories)
    out = {}
    for name, flag in (("All", None), ("Things", True), ("Stuff", False)):
        out[name], per_class = stat.average(categories, flag)
        if name == "All":
            out["per_class"] = per_class
    return out


def pq_compute(gt_json, pred_json, gt_folder, pred_folder):
    """File-based form with panopticapi's signature: json files in COCO panoptic format + folders of id pngs."""
    import json
    import os

    from PIL import Image

    from ..data.pseudo_panoptic import rgb2id

    gt, pred = json.load(open(gt_json)), json.load(open(pred_json))
    categories = {c["id"]: c for c in gt["categories"]}
    pred_by_image = {a["image_id"]: a for a in pred["annotations"]}

    def samples():
        for ga in gt["annotations"]:
            if ga["image_id"] not in pred_by_image:
                raise KeyError("no prediction for the image with id %r" % ga["image_id"])
            pa = pred_by_image[ga["image_id"]]
            yield (rgb2id(np.asarray(Image.open(os.path.join(gt_folder, ga["file_name"])))), ga["segments_info"],
                   rgb2id(np.asarray(Image.open(os.path.join(pred_folder, pa["file_name"])))), pa["segments_info"])

    return pq_compute_arrays(samples(), categories)
