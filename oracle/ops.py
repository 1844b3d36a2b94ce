"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only as the
checker.  The product path (u2seg_amd/) never imports, links or executes anything under oracle/.

Per-op CPU restatements (plain torch on CPU + the C file roi_ops.c) of the reference's hot-path arithmetic.
Every function cites the reference file:line it follows (paths relative to the reference root).  The oracle is
pinned against fixtures generated from the reference itself (tests/golden/make_fixtures.py, run in the build
container where /root/reference exists) and against the known answers of the reference's own unit tests.
"""
import ctypes
import math
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_clib = None


def clib():
    global _clib
    if _clib is None:
        path = os.path.join(_HERE, "liboracle_roi.so")
        if not os.path.exists(path):
            import subprocess

            subprocess.check_call(["make", "-C", _HERE, "-s"])
        lib = ctypes.CDLL(path)
        fp, ip = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)
        lib.oracle_roi_align_fwd.argtypes = [fp, fp, fp] + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        lib.oracle_roi_align_bwd.argtypes = [fp, fp, fp] + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        lib.oracle_nms.argtypes = [fp, fp, ip, ctypes.c_int, ctypes.c_float, ip]
        lib.oracle_nms.restype = ctypes.c_int
        _clib = lib
    return _clib


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


# ---------------------------------------------------------------------------------------------
# ROIAlign (torchvision.ops.roi_align; call sites detectron2/layers/roi_align.py:58-65)
# ---------------------------------------------------------------------------------------------
class _RoiAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, out_size, scale, sampling_ratio, aligned):
        f = np.ascontiguousarray(feat.detach().numpy(), dtype=np.float32)
        r = np.ascontiguousarray(rois.detach().numpy(), dtype=np.float32)
        n, c, h, w = f.shape
        out = np.zeros((r.shape[0], c, out_size, out_size), dtype=np.float32)
        clib().oracle_roi_align_fwd(_fptr(f), _fptr(r), _fptr(out), r.shape[0], c, h, w, out_size, out_size, float(scale),
                                    int(sampling_ratio), int(aligned))
        ctx.save_for_backward(rois)
        ctx.cfg = (f.shape, out_size, scale, sampling_ratio, aligned)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, dout):
        (rois,) = ctx.saved_tensors
        shape, out_size, scale, sampling_ratio, aligned = ctx.cfg
        d = np.ascontiguousarray(dout.numpy(), dtype=np.float32)
        r = np.ascontiguousarray(rois.numpy(), dtype=np.float32)
        g = np.zeros(shape, dtype=np.float32)
        clib().oracle_roi_align_bwd(_fptr(d), _fptr(r), _fptr(g), r.shape[0], shape[1], shape[2], shape[3], out_size,
                                    out_size, float(scale), int(sampling_ratio), int(aligned))
        return torch.from_numpy(g), None, None, None, None, None


def roi_align(feat_nchw, rois, out_size, scale, sampling_ratio=0, aligned=True):
    return _RoiAlignFn.apply(feat_nchw, rois, out_size, scale, sampling_ratio, aligned)


def nms(boxes, scores, thr, groups=None):
    """torchvision nms / batched_nms (detectron2/layers/nms.py:5-20): kept original indices by descending score."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    b = np.ascontiguousarray(boxes.detach().numpy(), dtype=np.float32)
    s = np.ascontiguousarray(scores.detach().float().numpy(), dtype=np.float32)
    keep = np.zeros(n, dtype=np.int32)
    gp = None
    if groups is not None:
        g = np.ascontiguousarray(groups.detach().numpy().astype(np.int32))
        gp = g.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
    nk = clib().oracle_nms(_fptr(b), _fptr(s), gp, n, float(thr), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return torch.from_numpy(keep[:nk].astype(np.int64))


# ---------------------------------------------------------------------------------------------
# Boxes / anchors / matching / sampling
# ---------------------------------------------------------------------------------------------
def pairwise_iou(b1, b2):
    """detectron2/structures/boxes.py:312-358."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype))


def matcher(mq, thresholds, labels, allow_low_quality_matches):
    """detectron2/modeling/matcher.py:62-127.  mq: [M gt, N candidates] -> (matches [N], match_labels [N] int8)."""
    thr = [-float("inf")] + list(thresholds) + [float("inf")]
    if mq.numel() == 0:
        default_matches = mq.new_full((mq.size(1),), 0, dtype=torch.int64)
        default_labels = mq.new_full((mq.size(1),), labels[0], dtype=torch.int8)
        return default_matches, default_labels
    matched_vals, matches = mq.max(dim=0)
    match_labels = matches.new_full(matches.size(), 1, dtype=torch.int8)
    for l, low, high in zip(labels, thr[:-1], thr[1:]):
        low_high = (matched_vals >= low) & (matched_vals < high)
        match_labels[low_high] = l
    if allow_low_quality_matches:
        highest, _ = mq.max(dim=1)
        _, pred_inds = torch.nonzero(mq == highest[:, None], as_tuple=True)
        match_labels[pred_inds] = 1
    return matches, match_labels


def subsample_labels(labels, num_samples, positive_fraction, bg_label, perm_fn=None):
    """detectron2/modeling/sampling.py:9-54 (perm_fn(n) replaces torch.randperm for injected permutations)."""
    positive = torch.nonzero((labels != -1) & (labels != bg_label), as_tuple=True)[0]
    negative = torch.nonzero(labels == bg_label, as_tuple=True)[0]
    num_pos = int(num_samples * positive_fraction)
    num_pos = min(positive.numel(), num_pos)
    num_neg = num_samples - num_pos
    num_neg = min(negative.numel(), num_neg)
    rp = perm_fn if perm_fn is not None else (lambda n: torch.randperm(n))
    perm1 = rp(positive.numel())[:num_pos]
    perm2 = rp(negative.numel())[:num_neg]
    return positive[perm1], negative[perm2]


def subsample_labels_keyed(labels, num_samples, positive_fraction, bg_label, keys):
    """detectron2/modeling/sampling.py:9-54 with the two ``torch.randperm`` draws replaced by the permutations a vector of
    per-candidate keys induces: ``randperm(P) := argsort(keys[positive])`` and ``randperm(N) := argsort(keys[negative])``
    (stable, so equal keys keep index order).  Everything else is subsample_labels above, line for line."""
    positive = torch.nonzero((labels != -1) & (labels != bg_label), as_tuple=True)[0]
    negative = torch.nonzero(labels == bg_label, as_tuple=True)[0]
    queue = [torch.argsort(keys[positive], stable=True), torch.argsort(keys[negative], stable=True)]
    sizes = [positive.numel(), negative.numel()]

    def perm_fn(n):
        assert n == sizes.pop(0)
        return queue.pop(0)

    return subsample_labels(labels, num_samples, positive_fraction, bg_label, perm_fn)


def generate_cell_anchors(sizes, aspect_ratios):
    """detectron2/modeling/anchor_generator.py:181-216."""
    anchors = []
    for size in sizes:
        area = size ** 2.0
        for ar in aspect_ratios:
            w = math.sqrt(area / ar)
            h = ar * w
            anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(anchors)


def grid_anchors(grid_sizes, strides, cell_anchors, offset=0.0):
    """detectron2/modeling/anchor_generator.py:39-55,165-179: (y, x, cell) order, x fastest."""
    out = []
    for (gh, gw), stride, base in zip(grid_sizes, strides, cell_anchors):
        sx = torch.arange(offset * stride, gw * stride, step=stride, dtype=torch.float32)
        sy = torch.arange(offset * stride, gh * stride, step=stride, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        shifts = torch.stack((xx, yy, xx, yy), dim=1)
        out.append((shifts.view(-1, 1, 4) + base.float().view(1, -1, 4)).reshape(-1, 4))
    return out


SCALE_CLAMP = math.log(1000.0 / 16)


def get_deltas(src, tgt, weights):
    """detectron2/modeling/box_regression.py:43-76."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    scx, scy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tcx, tcy = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tcx - scx) / sw, wy * (tcy - scy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)


def apply_deltas(deltas, boxes, weights, clamp=SCALE_CLAMP):
    """detectron2/modeling/box_regression.py:78-116 (always fp32)."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw, dh = deltas[:, 2::4] / ww, deltas[:, 3::4] / wh
    dw, dh = torch.clamp(dw, max=clamp), torch.clamp(dh, max=clamp)
    pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
    pw, ph = torch.exp(dw) * w[:, None], torch.exp(dh) * h[:, None]
    out = torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1)
    return out.reshape(deltas.shape)


def clip_boxes(boxes, size):
    """detectron2/structures/boxes.py:172-181."""
    h, w = size
    return torch.stack((boxes[:, 0].clamp(0, w), boxes[:, 1].clamp(0, h), boxes[:, 2].clamp(0, w), boxes[:, 3].clamp(0, h)), dim=-1)


def nonempty(boxes, threshold=0.0):
    return ((boxes[:, 2] - boxes[:, 0]) > threshold) & ((boxes[:, 3] - boxes[:, 1]) > threshold)


def assign_boxes_to_levels(boxes, min_level, max_level, canonical_box_size=224, canonical_level=4):
    """detectron2/modeling/poolers.py:23-59."""
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sizes = torch.sqrt(area)
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_pool_multilevel(feats, box_lists, out_size, scales):
    """ROIPooler.forward (detectron2/modeling/poolers.py:206-263) with ROIAlignV2."""
    boxes = torch.cat(box_lists, dim=0)
    idx = torch.cat([torch.full((len(b),), i, dtype=torch.float32) for i, b in enumerate(box_lists)])
    rois = torch.cat([idx[:, None], boxes], dim=1)
    min_level, max_level = int(-math.log2(scales[0])), int(-math.log2(scales[-1]))
    if len(feats) == 1:
        return roi_align(feats[0], rois, out_size, scales[0])
    lv = assign_boxes_to_levels(boxes, min_level, max_level)
    c = feats[0].shape[1]
    out = torch.zeros((rois.shape[0], c, out_size, out_size), dtype=feats[0].dtype)
    for level, (f, s) in enumerate(zip(feats, scales)):
        inds = torch.nonzero(lv == level, as_tuple=True)[0]
        if inds.numel():
            out = out.index_put((inds,), roi_align(f, rois[inds], out_size, s))
    return out


def crop_and_resize_masks(masks_bool, boxes, mask_size):
    """BitMasks.crop_and_resize (detectron2/structures/masks.py:191-218)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0, mask_size, mask_size), dtype=torch.bool)
    rois = torch.cat([torch.arange(n, dtype=torch.float32)[:, None], boxes.float()], dim=1)
    out = roi_align(masks_bool[:, None].float(), rois, mask_size, 1.0)
    return out.squeeze(1) >= 0.5


# ---------------------------------------------------------------------------------------------
# k-means (u2seg/Instance_Clustering/shared/utils/nn_utils.py:304-379, plain-torch branch :325-329)
# ---------------------------------------------------------------------------------------------
def kmeans_assign(x, c, chunk=4096):
    """cl = argmin_j sum_d (x_id - c_jd)^2, row-chunked so chunk*K*D floats fit in memory."""
    out = torch.empty(x.shape[0], dtype=torch.int64)
    for s in range(0, x.shape[0], chunk):
        d = ((x[s : s + chunk, None, :] - c[None, :, :]) ** 2).sum(-1)
        out[s : s + chunk] = d.argmin(dim=1)
    return out


def kmeans_update(x, cl, k):
    d = x.shape[1]
    c = torch.zeros((k, d), dtype=x.dtype)
    c.scatter_add_(0, cl[:, None].repeat(1, d), x)
    ncl = torch.bincount(cl, minlength=k).type_as(c).view(k, 1)
    c /= ncl
    return c, ncl.view(-1)


def kmeans(x, init_idx, niter):
    """Lloyd iterations from explicit initial indices (the reference draws them with randperm(N)[:K], :337-340)."""
    c = x[init_idx].clone()
    cl = None
    for _ in range(niter):
        cl = kmeans_assign(x, c)
        c, _ = kmeans_update(x, cl, c.shape[0])
    return cl, c


# ---------------------------------------------------------------------------------------------
# kNN density (u2seg/Instance_Clustering/shared/utils/nn_utils.py:204-299)
# ---------------------------------------------------------------------------------------------
def knn(x_train, x_test, k=20, chunk=512):
    """nn_utils.py:204-227.  D_ij = sum_d (x_test_id - x_train_jd)^2 in fp32 in the difference form the reference
    writes (:210-214), then pykeops' Kmin_argKmin(K, dim=1) (:216; pykeops 2.x is not in this image - its documented
    semantics restated: the K smallest values of each row in ascending order and their column indices; equal values are
    listed smaller column first, which pykeops leaves open).  Returns (ind_knn, d_knn) like the reference."""
    ind = torch.empty((x_test.shape[0], k), dtype=torch.int64)
    dk = torch.empty((x_test.shape[0], k), dtype=torch.float32)
    for s in range(0, x_test.shape[0], chunk):
        d = ((x_test[s : s + chunk, None, :] - x_train[None, :, :]) ** 2).sum(-1)
        v, i = torch.sort(d, dim=1, stable=True)
        dk[s : s + chunk], ind[s : s + chunk] = v[:, :k], i[:, :k]
    return ind, dk


def partitioned_knn(feats, k=20, partitions_size=130000):
    """nn_utils.py:230-266: every (train partition, test partition) pair contributes K candidates per test row, offset
    to global row numbers (:252-253); the K best of the partitions * K candidates are picked by an argsort (:259-266).
    Returns (d_knns, ind_knns)."""
    n = feats.shape[0]
    parts = -(-n // partitions_size)
    ind_all = torch.zeros((n, parts * k), dtype=torch.int64)
    d_all = torch.zeros((n, parts * k), dtype=torch.float32)
    for i in range(parts):
        tr = feats[i * partitions_size : (i + 1) * partitions_size]
        for j in range(parts):
            te = feats[j * partitions_size : (j + 1) * partitions_size]
            ind, d = knn(tr, te, k)
            ind_all[j * partitions_size : (j + 1) * partitions_size, i * k : (i + 1) * k] = i * partitions_size + ind
            d_all[j * partitions_size : (j + 1) * partitions_size, i * k : (i + 1) * k] = d
    sel = d_all.argsort(dim=1)[:, :k]
    return torch.gather(d_all, 1, sel), torch.gather(ind_all, 1, sel)
