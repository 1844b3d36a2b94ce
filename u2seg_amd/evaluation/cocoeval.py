"""COCO detection metrics (AP / AR over IoU thresholds, area ranges and detection budgets) without pycocotools.

The reference evaluates through pycocotools' COCOeval or its own C++ rewrite of the two heavy stages
(detectron2/evaluation/fast_eval_api.py:17-121 -> detectron2/layers/csrc/cocoeval/cocoeval.cpp: EvaluateImages :139-190 with
MatchDetectionsToGroundTruth :56-137, Accumulate :365-507 with ComputePrecisionRecallCurve :275-363).  This module restates
the whole pipeline in numpy - preparation and IoU as cocoapi documents them, matching and accumulation as that C++ does them -
and tests/test_evaluation.py compares its per-threshold precision / recall / score tables with the output of the reference's
C++ on the same instances.

Stages: prepare (group ground truth and detections by (image, category); a crowd region is "ignore"; a detection's area is
its box area) -> IoU of the score-sorted detections with the ground truth (intersection over the detection's area for
crowd regions) -> per (category, area range, image) greedy matching at every IoU threshold -> per (category, area range,
detection budget) precision / recall curves over all images -> the twelve summary numbers."""
import numpy as np

AREA_RANGES = ((0 ** 2, 1e5 ** 2), (0 ** 2, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e5 ** 2))
AREA_LABELS = ("all", "small", "medium", "large")


class Params:
    def __init__(self, img_ids, cat_ids, max_dets=(1, 10, 100)):
        self.imgIds, self.catIds = list(img_ids), list(cat_ids)
        self.iouThrs = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
        self.recThrs = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
        self.maxDets = list(max_dets)
        self.areaRng = [list(r) for r in AREA_RANGES]
        self.areaRngLbl = list(AREA_LABELS)
        self.useCats = 1


def box_ious(dt_boxes, gt_boxes, gt_crowd):
    """[D, G] IoU of xywh boxes; for a crowd ground truth the union is the detection's own area (cocoapi bbIou)."""
    d = np.asarray(dt_boxes, dtype=np.float64).reshape(-1, 4)
    g = np.asarray(gt_boxes, dtype=np.float64).reshape(-1, 4)
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    w = np.minimum(d[:, None, 0] + d[:, None, 2], g[None, :, 0] + g[None, :, 2]) - np.maximum(d[:, None, 0], g[None, :, 0])
    h = np.minimum(d[:, None, 1] + d[:, None, 3], g[None, :, 1] + g[None, :, 3]) - np.maximum(d[:, None, 1], g[None, :, 1])
    inter = np.where((w <= 0) | (h <= 0), 0.0, w * h)
    area_d, area_g = (d[:, 2] * d[:, 3])[:, None], (g[:, 2] * g[:, 3])[None, :]
    union = np.where(np.asarray(gt_crowd, dtype=bool)[None, :], area_d, area_d + area_g - inter)
    return inter / union


def prepare(gt_annotations, results, params):
    """{(image id, category id): [instances]} for ground truth and detections.  Instances are dicts with id, bbox, area,
    score (detections), iscrowd and ignore (ground truth: a crowd region is ignored, cocoeval.py `_prepare`)."""
    imgs, cats = set(params.imgIds), set(params.catIds)
    gts, dts = {}, {}
    for ann in gt_annotations:
        if ann["image_id"] in imgs and ann["category_id"] in cats:
            crowd = int(ann.get("iscrowd", 0))
            area = ann["area"] if "area" in ann else ann["bbox"][2] * ann["bbox"][3]
            gts.setdefault((ann["image_id"], ann["category_id"]), []).append(
                {"id": ann["id"], "bbox": ann["bbox"], "area": area, "iscrowd": crowd, "ignore": crowd})
    for k, res in enumerate(results):  # loadRes: ids 1.., area = box area, not a crowd
        if res["image_id"] in imgs and res["category_id"] in cats:
            dts.setdefault((res["image_id"], res["category_id"]), []).append(
                {"id": k + 1, "bbox": res["bbox"], "area": res["bbox"][2] * res["bbox"][3], "score": res["score"],
                 "iscrowd": 0, "ignore": 0})
    return gts, dts


def compute_ious(gts, dts, params):
    """{(image, category): [D', G] array} with the detections in descending score order, at most maxDets[-1] of them."""
    out = {}
    for img in params.imgIds:
        for cat in params.catIds:
            gt, dt = gts.get((img, cat), []), dts.get((img, cat), [])
            if not gt and not dt:
                out[img, cat] = []
                continue
            order = np.argsort([-d["score"] for d in dt], kind="mergesort")[: params.maxDets[-1]]
            out[img, cat] = box_ious([dt[i]["bbox"] for i in order], [g["bbox"] for g in gt], [g["iscrowd"] for g in gt])
    return out


def _match_image(gt, dt, ious, area_range, iou_thrs, max_det):
    """One (image, category, area range): detections in score order are matched greedily, at every IoU threshold, to the
    not yet matched ground truth of highest IoU (crowd regions can be matched repeatedly; a match with a regular instance
    is never given up for an ignored one).  Returns (matched gt id or 0 [T, D], detection ignored [T, D], scores [D],
    gt ignored [G] in the order used)."""
    d_order = np.argsort([-d["score"] for d in dt], kind="mergesort")[:max_det]
    g_ignore = np.array([bool(g["ignore"]) or g["area"] < area_range[0] or g["area"] > area_range[1] for g in gt], dtype=bool)
    g_order = np.argsort(g_ignore.astype(np.int64), kind="mergesort")
    g_ign = g_ignore[g_order]
    T, D, G = len(iou_thrs), len(d_order), len(g_order)
    dt_match = np.zeros((T, D), dtype=np.int64)
    dt_ignore = np.zeros((T, D), dtype=bool)
    gt_taken = np.zeros((T, G), dtype=bool)
    crowd = np.array([bool(gt[j]["iscrowd"]) for j in g_order], dtype=bool)
    for t, thr in enumerate(iou_thrs):
        for d in range(D):
            best, match = min(thr, 1 - 1e-10), -1
            for g in range(G):
                if gt_taken[t, g] and not crowd[g]:
                    continue
                if match >= 0 and not g_ign[match] and g_ign[g]:
                    break
                v = ious[d][g_order[g]]
                if v >= best:
                    best, match = v, g
            if match >= 0:
                dt_ignore[t, d] = g_ign[match]
                dt_match[t, d] = gt[g_order[match]]["id"]
                gt_taken[t, match] = True
            det = dt[d_order[d]]
            outside = det["area"] < area_range[0] or det["area"] > area_range[1]
            dt_ignore[t, d] = dt_ignore[t, d] or (dt_match[t, d] == 0 and outside)
    scores = np.array([dt[i]["score"] for i in d_order], dtype=np.float64)
    return dt_match, dt_ignore, scores, g_ign


def evaluate_images(gts, dts, ious, params):
    """[category][area range][image] -> match record (or None when the image has neither gt nor detections there)."""
    out = []
    for cat in params.catIds:
        per_area = []
        for rng in params.areaRng:
            per_img = []
            for img in params.imgIds:
                gt, dt = gts.get((img, cat), []), dts.get((img, cat), [])
                per_img.append(_match_image(gt, dt, ious[img, cat], rng, params.iouThrs, params.maxDets[-1]))
            per_area.append(per_img)
        out.append(per_area)
    return out


def accumulate(evaluations, params):
    """precision / scores [T, R, K, A, M] and recall [T, K, A, M]; -1 where a (category, area range) has no valid gt."""
    T, R, K, A, M = len(params.iouThrs), len(params.recThrs), len(params.catIds), len(params.areaRng), len(params.maxDets)
    precision = -np.ones((T, R, K, A, M))
    scores = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for k in range(K):
        for a in range(A):
            records = evaluations[k][a]
            n_valid = int(sum((~rec[3]).sum() for rec in records))
            if n_valid == 0:
                continue
            for m, budget in enumerate(params.maxDets):
                sc = np.concatenate([rec[2][:budget] for rec in records])
                order = np.argsort(-sc, kind="mergesort")
                sc_sorted = sc[order]
                matched = np.concatenate([rec[0][:, :budget] for rec in records], axis=1)[:, order]
                ignored = np.concatenate([rec[1][:, :budget] for rec in records], axis=1)[:, order]
                tp = np.cumsum((matched > 0) & ~ignored, axis=1)
                fp = np.cumsum((matched == 0) & ~ignored, axis=1)
                for t in range(T):
                    nd = tp.shape[1]
                    rc = tp[t] / n_valid
                    valid = tp[t] + fp[t]
                    pr = np.where(valid > 0, tp[t] / np.maximum(valid, 1), 0.0)
                    recall[t, k, a, m] = rc[-1] if nd else 0
                    for i in range(nd - 1, 0, -1):  # precision envelope
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    idx = np.searchsorted(rc, params.recThrs, side="left")
                    ok = idx < nd
                    precision[t, :, k, a, m] = np.where(ok, pr[np.minimum(idx, max(nd - 1, 0))] if nd else 0.0, 0.0)
                    scores[t, :, k, a, m] = np.where(ok, sc_sorted[np.minimum(idx, max(nd - 1, 0))] if nd else 0.0, 0.0)
    return {"precision": precision, "recall": recall, "scores": scores, "counts": [T, R, K, A, M]}


def summarize(acc, params):
    """The 12 numbers of COCOeval.summarize() for boxes / masks: AP, AP50, AP75, APs, APm, APl, AR@1, AR@10, AR@100,
    ARs, ARm, ARl (mean over the entries that are not -1; -1 when there is none)."""
    def pick(ap, iou=None, area="all", max_det=100):
        a, m = params.areaRngLbl.index(area), params.maxDets.index(max_det)
        s = acc["precision"][:, :, :, a, m] if ap else acc["recall"][:, :, a, m]
        if iou is not None:
            s = s[np.where(np.isclose(params.iouThrs, iou))[0]]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0

    last = params.maxDets[-1]
    return [pick(True, max_det=last), pick(True, 0.5, max_det=last), pick(True, 0.75, max_det=last),
            pick(True, area="small", max_det=last), pick(True, area="medium", max_det=last), pick(True, area="large", max_det=last),
            pick(False, max_det=params.maxDets[0]), pick(False, max_det=params.maxDets[1]), pick(False, max_det=last),
            pick(False, area="small", max_det=last), pick(False, area="medium", max_det=last), pick(False, area="large", max_det=last)]


STAT_NAMES = ("AP", "AP50", "AP75", "APs", "APm", "APl", "AR1", "AR10", "AR100", "ARs", "ARm", "ARl")


def evaluate_bbox(gt_dataset, results, img_ids=None, max_dets=(1, 10, 100)):
    """gt_dataset: the COCO json dict (images, annotations, categories); results: COCO result dicts with xywh boxes.
    Returns {"stats": {name: value in [0, 1] or -1}, "precision", "recall", "params"}."""
    imgs = sorted(im["id"] for im in gt_dataset["images"]) if img_ids is None else sorted(set(img_ids))
    cats = sorted(c["id"] for c in gt_dataset["categories"])
    params = Params(imgs, cats, max_dets)
    gts, dts = prepare(gt_dataset["annotations"], results, params)
    ious = compute_ious(gts, dts, params)
    acc = accumulate(evaluate_images(gts, dts, ious, params), params)
    return {"stats": dict(zip(STAT_NAMES, summarize(acc, params))), "precision": acc["precision"], "recall": acc["recall"],
            "scores": acc["scores"], "params": params}
