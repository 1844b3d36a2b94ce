"""Inference-only tails: box filtering + per-class NMS, mask pasting, output rescaling and the panoptic merge
(detectron2/modeling/roi_heads/fast_rcnn.py:46-171, layers/mask_ops.py:17-147, modeling/postprocessing.py:9-100,
meta_arch/panoptic_fpn.py:184-269).  NMS runs in the HIP kernel; the remaining index plumbing is device-side torch."""
import torch

from ..layers import functional as F
from ..structures import Boxes, Instances


def nms_single(boxes, scores, groups, thr, topk):
    """batched_nms for one image: returns kept indices sorted by descending score."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True)[1]
    sb = boxes[order][None].contiguous()
    sg = groups[order].to(torch.int32)[None].contiguous()
    cnt = torch.tensor([n], dtype=torch.int32, device=boxes.device)
    max_keep = n if topk < 0 else min(n, topk)
    keep, nkeep = F.batched_nms(sb, sg, cnt, thr, max_keep)
    return order[keep[0, : int(nkeep[0])].long()]


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk_per_image):
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    if not bool(valid.all()):
        boxes, scores = boxes[valid], scores[valid]
    scores = scores[:, :-1]
    num_bbox_reg_classes = boxes.shape[1] // 4
    b = Boxes(boxes.reshape(-1, 4))
    b.clip(image_shape)
    boxes = b.tensor.view(-1, num_bbox_reg_classes, 4)
    filter_mask = scores > score_thresh
    filter_inds = filter_mask.nonzero()
    boxes = boxes[filter_inds[:, 0], 0] if num_bbox_reg_classes == 1 else boxes[filter_mask]
    scores = scores[filter_mask]
    keep = nms_single(boxes, scores, filter_inds[:, 1], nms_thresh, topk_per_image)
    boxes, scores, filter_inds = boxes[keep], scores[keep], filter_inds[keep]
    result = Instances(image_shape)
    result.pred_boxes = Boxes(boxes)
    result.scores = scores
    result.pred_classes = filter_inds[:, 1]
    return result, filter_inds[:, 0]


def fast_rcnn_inference(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image):
    out = [fast_rcnn_inference_single_image(b, s, shp, score_thresh, nms_thresh, topk_per_image)
           for s, b, shp in zip(scores, boxes, image_shapes)]
    return [x[0] for x in out], [x[1] for x in out]


def paste_masks_in_image(masks, boxes, image_shape, threshold=0.5):
    """mask_ops.py:17-147 (GPU branch: full-image grid_sample, aligned at pixel centres)."""
    n = masks.shape[0]
    img_h, img_w = image_shape
    if n == 0:
        return masks.new_empty((0, img_h, img_w), dtype=torch.bool)
    device = masks.device
    out = torch.empty((n, img_h, img_w), dtype=torch.bool, device=device)
    chunk = max(1, int((1 << 28) // max(img_h * img_w, 1)))
    for s in range(0, n, chunk):
        m = masks[s : s + chunk, None].float()
        bx = boxes[s : s + chunk]
        x0, y0, x1, y1 = torch.split(bx, 1, dim=1)
        img_y = torch.arange(0, img_h, device=device, dtype=torch.float32) + 0.5
        img_x = torch.arange(0, img_w, device=device, dtype=torch.float32) + 0.5
        img_y = (img_y - y0) / (y1 - y0) * 2 - 1
        img_x = (img_x - x0) / (x1 - x0) * 2 - 1
        gx = img_x[:, None, :].expand(m.shape[0], img_y.size(1), img_x.size(1))
        gy = img_y[:, :, None].expand(m.shape[0], img_y.size(1), img_x.size(1))
        grid = torch.stack([gx, gy], dim=3)
        img = torch.nn.functional.grid_sample(m, grid, align_corners=False)
        out[s : s + chunk] = img[:, 0] >= threshold
    return out


def detector_postprocess(results, output_height, output_width, mask_threshold=0.5):
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    output_boxes = results.pred_boxes if results.has("pred_boxes") else results.proposal_boxes
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    results = results[output_boxes.nonempty()]
    if results.has("pred_masks"):
        results.pred_masks = paste_masks_in_image(results.pred_masks[:, 0, :, :], results.pred_boxes.tensor,
                                                  (output_height, output_width), mask_threshold)
    return results


def sem_seg_postprocess(result, img_size, output_height, output_width):
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return torch.nn.functional.interpolate(result, size=(output_height, output_width), mode="bilinear", align_corners=False)[0]


def combine_semantic_and_instance_outputs(instance_results, semantic_results, overlap_threshold, stuff_area_thresh,
                                          instances_score_thresh):
    """panoptic_fpn.py:184-269.  Areas are reduced on the device in one batch and read back once; the greedy
    paste order (descending score) and every integer decision follow the reference."""
    panoptic_seg = torch.zeros_like(semantic_results, dtype=torch.int32)
    sorted_inds = torch.argsort(-instance_results.scores)
    current_segment_id = 0
    segments_info = []
    instance_masks = instance_results.pred_masks.to(dtype=torch.bool, device=panoptic_seg.device)
    scores = instance_results.scores[sorted_inds].tolist()
    classes = instance_results.pred_classes[sorted_inds].tolist()
    areas = instance_masks.flatten(1).sum(1)[sorted_inds].tolist() if len(scores) else []
    for rank, inst_id in enumerate(sorted_inds.tolist()):
        score = scores[rank]
        if score < instances_score_thresh:
            break
        mask = instance_masks[inst_id]
        mask_area = areas[rank]
        if mask_area == 0:
            continue
        intersect_area = int((mask & (panoptic_seg > 0)).sum())
        if intersect_area * 1.0 / mask_area > overlap_threshold:
            continue
        if intersect_area > 0:
            mask = mask & (panoptic_seg == 0)
        current_segment_id += 1
        panoptic_seg[mask] = current_segment_id
        segments_info.append({"id": current_segment_id, "isthing": True, "score": score, "category_id": classes[rank],
                              "instance_id": inst_id})
    semantic_labels = torch.unique(semantic_results).cpu().tolist()
    for semantic_label in semantic_labels:
        if semantic_label == 0:
            continue
        mask = (semantic_results == semantic_label) & (panoptic_seg == 0)
        mask_area = int(mask.sum())
        if mask_area < stuff_area_thresh:
            continue
        current_segment_id += 1
        panoptic_seg[mask] = current_segment_id
        segments_info.append({"id": current_segment_id, "isthing": False, "category_id": semantic_label, "area": mask_area})
    return panoptic_seg, segments_info
