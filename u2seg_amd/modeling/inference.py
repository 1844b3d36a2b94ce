"""Inference-only tails: box filtering + per-class NMS, mask pasting, output rescaling and the panoptic merge
(detectron2/modeling/roi_heads/fast_rcnn.py:46-171, layers/mask_ops.py:17-147, modeling/postprocessing.py:9-100,
meta_arch/panoptic_fpn.py:184-269).  NMS runs in the HIP kernel; the remaining index plumbing is device-side torch."""
import ctypes

import torch

from .. import _hip
from ..layers import functional as F
from ..structures import Boxes, Instances
from .batched import device_constant, device_upload


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    """fast_rcnn.py:117-171 for one image (the batched routine below with a batch of one)."""
    res, kept = fast_rcnn_inference([boxes], [scores], [image_shape], score_thresh, nms_thresh, topk_per_image)
    return res[0], kept[0]


def fast_rcnn_inference(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image):
    """fast_rcnn.py:46-171 for the whole batch at once: score filter, per-class NMS and top-k with three host
    synchronisations per batch (candidate counts, the candidate list, keep counts) instead of four per image.
    boxes: per image [R_i, 4] (class agnostic) or [R_i, K*4]; scores: per image [R_i, K+1]; or both stacked, [B, R, .]."""
    nimg = len(boxes)
    if nimg == 0:
        return [], []
    dev = boxes[0].device
    k = scores[0].shape[1] - 1
    nreg = boxes[0].shape[1] // 4
    counts_r = [b.shape[0] for b in boxes]
    rmax = max(max(counts_r), 1)
    if isinstance(boxes, torch.Tensor) and isinstance(scores, torch.Tensor) and counts_r[0] > 0:
        bx, sc = boxes, scores  # already stacked [B, R, .]: no copy
        rvalid = None
    elif len(set(counts_r)) == 1 and counts_r[0] > 0:
        bx, sc = torch.stack(list(boxes)), torch.stack(list(scores))
        rvalid = None
    else:  # ragged: pad with rows that never pass the score filter
        bx = torch.zeros((nimg, rmax, boxes[0].shape[1]), dtype=boxes[0].dtype, device=dev)
        sc = torch.full((nimg, rmax, k + 1), -1.0, dtype=scores[0].dtype, device=dev)
        for i, (b_, s_) in enumerate(zip(boxes, scores)):
            bx[i, : b_.shape[0]], sc[i, : s_.shape[0]] = b_, s_
        rvalid = torch.arange(rmax, device=dev)[None] < device_upload(counts_r, torch.int64, dev)[:, None]
    valid = torch.isfinite(bx).all(dim=2) & torch.isfinite(sc).all(dim=2)
    if rvalid is not None:
        valid = valid & rvalid
    lim = device_constant([[[w, h] * 2] for h, w in image_shapes], torch.float32, dev)  # Boxes.clip per image
    bx = torch.minimum(bx.float().view(nimg, rmax, nreg, 4).clamp(min=0), lim[:, :, None, :])
    cand = (sc[..., :k] > score_thresh) & valid[..., None]  # [B, R, K]
    idx = cand.nonzero()  # sync 1; rows ordered by (image, roi, class) like the reference's per-image nonzero
    ncand = torch.bincount(idx[:, 0], minlength=nimg)  # (a sum over the [B, R * K] mask was 0.28 ms of a 32-image batch)
    cnt = ncand.tolist()  # sync 2
    total = idx.shape[0]
    c_img, c_roi, c_cls = idx[:, 0], idx[:, 1], idx[:, 2]
    c_scores = sc[c_img, c_roi, c_cls]
    c_boxes = bx[c_img, c_roi, 0] if nreg == 1 else bx[c_img, c_roi, c_cls]
    # per image: descending score, ties in candidate order (torch.sort(descending, stable) of the reference's NMS)
    o1 = torch.sort(c_scores, descending=True, stable=True)[1]
    o2 = torch.sort(c_img[o1], stable=True)[1]
    order = o1[o2]
    offs = [0]
    for c in cnt:
        offs.append(offs[-1] + c)
    nmax = max(max(cnt), 1)
    s_img = c_img[order]
    pos = torch.arange(total, device=dev) - device_upload(offs[:-1], torch.int64, dev)[s_img]
    pb = torch.zeros((nimg, nmax, 4), dtype=torch.float32, device=dev)
    pg = torch.zeros((nimg, nmax), dtype=torch.int32, device=dev)
    pb[s_img, pos] = c_boxes[order]
    pg[s_img, pos] = c_cls[order].to(torch.int32)
    max_keep = nmax if topk_per_image < 0 else min(nmax, topk_per_image)
    keep, nkeep = F.batched_nms(pb, pg, ncand.to(torch.int32), nms_thresh, max_keep)
    nk = nkeep.tolist()  # sync 3
    # the kept candidates of all images with one gather per field (per image: seven small launches, 220 per 32-image batch);
    # the per-image results are views of the gathered tensors
    live = torch.arange(keep.shape[1], device=dev)[None] < nkeep[:, None]
    sel_all = order[(device_upload(offs[:-1], torch.int64, dev)[:, None] + keep.long())[live]]  # image-major, kept order
    g_boxes, g_scores, g_cls, g_roi = c_boxes[sel_all], c_scores[sel_all], c_cls[sel_all], c_roi[sel_all]
    results, kept_rows = [], []
    for shape, b_, s_, c_, r_ in zip(image_shapes, g_boxes.split(nk), g_scores.split(nk), g_cls.split(nk), g_roi.split(nk)):
        res = Instances(shape)
        res.pred_boxes = Boxes(b_)
        res.scores = s_
        res.pred_classes = c_
        results.append(res)
        kept_rows.append(r_)
    return results, kept_rows


def paste_masks_in_image(masks, boxes, image_shape, threshold=0.5):
    """mask_ops.py:17-147 (GPU branch: full-image grid_sample, aligned at pixel centres) as one HIP kernel that samples
    and thresholds in place: the fp32 [n, H, W] sampled image of the reference is never formed."""
    n = masks.shape[0]
    img_h, img_w = image_shape
    if n == 0:
        return masks.new_empty((0, img_h, img_w), dtype=torch.bool)
    out = torch.empty((n, img_h, img_w), dtype=torch.bool, device=masks.device)
    _hip.call("u2_paste_masks", masks.float().contiguous(), boxes.float().contiguous(), out, n, masks.shape[-1], img_h, img_w,
              float(threshold))
    return out


def detector_postprocess_batch(results_list, sizes, mask_threshold=0.5):
    """postprocessing.py:9-74 for a batch: rescale + clip the boxes, drop empty ones, paste the masks.  The "is any box
    empty" question is answered for all images with one host synchronisation (the reference indexes with a boolean mask,
    i.e. synchronises, per image)."""
    if not results_list:
        return []
    # Boxes.scale + Boxes.clip + nonempty (structures/boxes.py:27-34,58-60) for the boxes of ALL images at once: the per-image
    # form was ~14 small launches per image (450 per 32-image batch); per-row scale factors and limits come from cached constants
    name = "pred_boxes" if results_list[0].has("pred_boxes") else "proposal_boxes"
    counts = [len(r) for r in results_list]
    tensors = [r.get(name).tensor for r in results_list]
    dev = tensors[0].device
    scl, lim = [], []
    for i, (results, (output_height, output_width)) in enumerate(zip(results_list, sizes)):
        sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
        scl.append([sx, sy, sx, sy])
        lim.append([output_width, output_height, output_width, output_height])
    total = sum(counts)
    if total:
        img_idx = torch.repeat_interleave(device_constant(list(range(len(counts))), torch.int64, dev),
                                          device_upload(counts, torch.int64, dev), output_size=total)
        t = torch.cat(tensors).float() * device_constant(scl, torch.float32, dev)[img_idx]
        finite_row = torch.isfinite(t).all(dim=1)
        t = torch.minimum(t.clamp(min=0), device_constant(lim, torch.float32, dev)[img_idx])
        keep_row = ((t[:, 2] - t[:, 0]) > 0.0) & ((t[:, 3] - t[:, 1]) > 0.0)
        bad = torch.zeros((len(results_list), 2), dtype=torch.int32, device=dev)
        bad.index_add_(0, img_idx, torch.stack([~keep_row, ~finite_row], dim=1).to(torch.int32))
        flags = bad.tolist()  # the one host synchronisation
        boxes_per_image, keep_per_image = t.split(counts), keep_row.split(counts)
    else:
        flags = [[0, 0]] * len(results_list)
        boxes_per_image = [x.float() for x in tensors]
        keep_per_image = [torch.ones(0, dtype=torch.bool, device=dev)] * len(results_list)
    out = []
    for results, (output_height, output_width), bx, keep, (n_empty, n_nonfinite) in zip(results_list, sizes, boxes_per_image,
                                                                                       keep_per_image, flags):
        assert not n_nonfinite, "Box tensor contains infinite or NaN!"
        results = Instances((output_height, output_width), **results.get_fields())
        results.set(name, Boxes(bx))
        if n_empty:
            results = results[keep]
        out.append(results)
    if out[0].has("pred_masks"):
        pasted = paste_masks_in_images([r.pred_masks[:, 0, :, :] for r in out], [r.pred_boxes.tensor for r in out],
                                       [r.image_size for r in out], mask_threshold)
        for r, m in zip(out, pasted):
            r.pred_masks = m
    return out


class _PasteImage(ctypes.Structure):
    """Mirror of U2PasteImage (include/u2seg_hip.h)."""

    _fields_ = [("first", ctypes.c_int), ("n", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
                ("out_offset", ctypes.c_longlong)]


def paste_masks_in_images(masks_list, boxes_list, image_shapes, threshold=0.5):
    """paste_masks_in_image for a batch of images with ONE kernel launch (the reference pastes per image,
    postprocessing.py:61-70): the masks [n_i, P, P] and boxes [n_i, 4] of all images are concatenated, every image gets its own
    canvas size; returns a list of bool [n_i, H_i, W_i] views of one allocation."""
    counts = [int(m.shape[0]) for m in masks_list]
    dev = masks_list[0].device
    descs = (_PasteImage * len(masks_list))()
    first, off = 0, 0
    for i, (n, (h, w)) in enumerate(zip(counts, image_shapes)):
        d = descs[i]
        d.first, d.n, d.H, d.W, d.out_offset = first, n, int(h), int(w), off
        first += n
        off += (n * int(h) * int(w) + 15) // 16 * 16
    flat = torch.empty(max(off, 16), dtype=torch.bool, device=dev)
    if first:
        p = int(masks_list[0].shape[-1])
        _hip.call("u2_paste_masks_batch", torch.cat(masks_list).float().contiguous(), torch.cat(boxes_list).float().contiguous(),
                  flat, descs, len(masks_list), p, float(threshold))
    out = []
    for i, (n, (h, w)) in enumerate(zip(counts, image_shapes)):
        o = descs[i].out_offset
        out.append(flat[o : o + n * int(h) * int(w)].view(n, int(h), int(w)))
    return out


def detector_postprocess(results, output_height, output_width, mask_threshold=0.5):
    return detector_postprocess_batch([results], [(output_height, output_width)], mask_threshold)[0]


def sem_seg_postprocess(result, img_size, output_height, output_width):
    result = result[:, : img_size[0], : img_size[1]]
    if (output_height, output_width) == tuple(img_size):
        return result  # bilinear resampling to the same size (align_corners=False) is the identity
    if result.is_cuda and result.dtype == torch.float32 and result.stride(2) == 1:
        # the cropped window is read in place (channel / row strides): no copy of the crop, one launch
        out = torch.empty((result.shape[0], output_height, output_width), dtype=torch.float32, device=result.device)
        _hip.call("u2_bilinear_resize_f32", result, out, result.shape[0], img_size[0], img_size[1], result.stride(0),
                  result.stride(1), output_height, output_width)
        return out
    result = result.expand(1, -1, -1, -1)
    return torch.nn.functional.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


class _PanopticImage(ctypes.Structure):
    """Mirror of U2PanopticImage (include/u2seg_hip.h)."""

    _fields_ = [("masks", ctypes.c_void_p), ("order", ctypes.c_void_p), ("scores_sorted", ctypes.c_void_p),
                ("boxes", ctypes.c_void_p), ("semantic", ctypes.c_void_p), ("panoptic", ctypes.c_void_p),
                ("inst_segment", ctypes.c_void_p), ("stuff_segment", ctypes.c_void_p), ("stuff_area", ctypes.c_void_p),
                ("K", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("num_sem", ctypes.c_int),
                ("sem_stride", ctypes.c_int)]


_NUM_SEM_SLOTS = 256  # semantic labels the merge kernel can number (PM_MAXSEM)


def combine_semantic_and_instance_outputs_batch(instance_results, semantic_results, overlap_threshold, stuff_area_thresh,
                                                instances_score_thresh, mask_res=0):
    """panoptic_fpn.py:184-269 for a list of images in one kernel launch and one device->host transfer.

    instance_results: list[Instances] with pred_masks [K, H, W] bool (pasted), scores, pred_classes (and pred_boxes, used
    with mask_res > 0 to bound the pixels each pasted mask can touch); semantic_results: list of [H, W] int64 argmax maps.
    Returns list[(panoptic_seg int32 [H, W], segments_info)] identical to the reference's per-image routine."""
    n = len(instance_results)
    if n == 0:
        return []
    dev = semantic_results[0].device
    descs = (_PanopticImage * n)()
    ks = [len(inst) for inst in instance_results]
    kmax, ktot = max(ks), sum(ks)
    # The instance order of ALL images with one sort (per image: negate + sort + casts + gather, ~8 launches x 32 images): the
    # scores go into a [n, kmax] matrix padded with -inf, torch.argsort(-scores) - the reference's expression, ties included
    # (the radix sort is stable) - runs along its rows; row i's first k_i entries are image i's order.
    if ktot:
        flat_scores = torch.cat([inst.scores.float() for inst in instance_results])
        rows = [i for i, k in enumerate(ks) for _ in range(k)]
        cols = [j for k in ks for j in range(k)]
        spad = torch.full((n, kmax), float("-inf"), dtype=torch.float32, device=dev)
        spad[device_upload(rows, torch.int64, dev), device_upload(cols, torch.int64, dev)] = flat_scores
        order2d = torch.argsort(-spad, dim=1, stable=True)  # equal scores keep their detection order, as the 1-D per-image sort does
        sorted2d = torch.gather(spad, 1, order2d).contiguous()
        order2d = order2d.to(torch.int32).contiguous()
        classes_all = torch.cat([inst.pred_classes for inst in instance_results]).to(torch.int32)
    else:
        order2d = torch.zeros((n, 1), dtype=torch.int32, device=dev)
        sorted2d = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        classes_all = torch.zeros(0, dtype=torch.int32, device=dev)
        kmax = 1
    out_all = torch.zeros(ktot + 2 * _NUM_SEM_SLOTS * n, dtype=torch.int32, device=dev)
    keep_alive, per_image = [], []
    opos = 0
    for i, (inst, sem) in enumerate(zip(instance_results, semantic_results)):
        h, w = sem.shape
        k = ks[i]
        # the label map is read in place when it is an int64 window of a wider map (the fused argmax of the padded batch)
        if not (sem.dtype == torch.int64 and sem.stride(1) == 1 and sem.stride(0) >= w):
            sem = sem.to(torch.int64).contiguous()
        masks = inst.pred_masks.to(device=dev)
        masks = (masks if masks.dtype in (torch.bool, torch.uint8) else masks > 0).contiguous()
        assert masks.shape == (k, h, w), (masks.shape, (k, h, w))
        boxes = inst.pred_boxes.tensor.float().contiguous() if (mask_res > 0 and inst.has("pred_boxes")) else None
        pan = torch.empty((h, w), dtype=torch.int32, device=dev)
        d = descs[i]
        d.masks = masks.data_ptr()
        d.order, d.scores_sorted = order2d.data_ptr() + 4 * kmax * i, sorted2d.data_ptr() + 4 * kmax * i
        d.boxes = boxes.data_ptr() if boxes is not None else None
        d.semantic, d.panoptic = sem.data_ptr(), pan.data_ptr()
        d.inst_segment = out_all.data_ptr() + 4 * opos
        d.stuff_segment = out_all.data_ptr() + 4 * (opos + k)
        d.stuff_area = out_all.data_ptr() + 4 * (opos + k + _NUM_SEM_SLOTS)
        d.K, d.H, d.W, d.num_sem, d.sem_stride = k, h, w, _NUM_SEM_SLOTS, sem.stride(0)
        opos += k + 2 * _NUM_SEM_SLOTS
        keep_alive.append((masks, boxes, sem))
        per_image.append((pan, k))
    _hip.call("u2_panoptic_merge", descs, n, float(overlap_threshold), int(stuff_area_thresh), float(instances_score_thresh),
              int(mask_res))
    # (after the launch) the one host synchronisation: segment ids / areas, the orders and the classes in one transfer
    flat = torch.cat([out_all, order2d.reshape(-1), classes_all]).tolist()
    score_rows = sorted2d.tolist() if ktot else [[] for _ in range(n)]
    o_base, c_base = out_all.numel(), out_all.numel() + order2d.numel()
    results, pos, cpos = [], 0, 0
    for i, (pan, k) in enumerate(per_image):
        inst_seg = flat[pos : pos + k]
        stuff_seg = flat[pos + k : pos + k + _NUM_SEM_SLOTS]
        stuff_area = flat[pos + k + _NUM_SEM_SLOTS : pos + k + 2 * _NUM_SEM_SLOTS]
        pos += k + 2 * _NUM_SEM_SLOTS
        order = flat[o_base + i * kmax : o_base + i * kmax + k]
        classes = flat[c_base + cpos : c_base + cpos + k]
        cpos += k
        scores = score_rows[i][:k]
        info = []
        for rank in range(k):
            if inst_seg[rank] > 0:
                inst_id = order[rank]
                info.append({"id": inst_seg[rank], "isthing": True, "score": scores[rank], "category_id": classes[inst_id],
                             "instance_id": inst_id})
        for label in range(1, _NUM_SEM_SLOTS):
            if stuff_seg[label] > 0:
                info.append({"id": stuff_seg[label], "isthing": False, "category_id": label, "area": stuff_area[label]})
        results.append((pan, info))
    return results


def combine_semantic_and_instance_outputs(instance_results, semantic_results, overlap_threshold, stuff_area_thresh,
                                          instances_score_thresh):
    """panoptic_fpn.py:184-269 (the batch routine above with one image)."""
    return combine_semantic_and_instance_outputs_batch([instance_results], [semantic_results], overlap_threshold,
                                                       stuff_area_thresh, instances_score_thresh)[0]
