"""ROI heads: FPN level assignment + ROIAlign pooler, FC box head, output layers, mask head and the
three-stage cascade (detectron2/modeling/poolers.py:23-263, roi_heads/box_head.py:26-118,
roi_heads/fast_rcnn.py:46-569, roi_heads/mask_head.py:33-298, roi_heads/roi_heads.py:46-846,
roi_heads/cascade_rcnn.py:20-299).  Per-image Python loops remain only for the index bookkeeping the
reference also does per image; all arithmetic on features runs in the HIP kernels."""
import math

import os

import torch
from torch import nn

from ..config import configurable
from ..layers import Conv2d, ConvTranspose2d, Linear, ShapeSpec, c2_msra_fill, c2_xavier_fill
from ..layers import functional as F
from ..structures import BitMasks, Boxes, Instances
from ..utils.registry import Registry
from .batched import BatchList, PaddedTargets, check_finite, device_constant, device_upload, image_index, proposals_from_list
from .sampling import subsample_labels

ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
ROI_MASK_HEAD_REGISTRY = Registry("ROI_MASK_HEAD")

_SCALE_CLAMP = math.log(1000.0 / 16)


# ---------------------------------------------------------------------------------------------
class ROIPooler(nn.Module):
    """poolers.py:114-263 with pooler_type ROIAlignV2 (aligned=True): one multi-level kernel launch."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type="ROIAlignV2", canonical_box_size=224,
                 canonical_level=4):
        super().__init__()
        assert pooler_type == "ROIAlignV2" and sampling_ratio == 0
        self.output_size = output_size if isinstance(output_size, int) else output_size[0]
        self.scales = tuple(scales)
        min_level = -(math.log2(scales[0]))
        max_level = -(math.log2(scales[-1]))
        assert math.isclose(min_level, int(min_level)) and math.isclose(max_level, int(max_level))
        self.min_level, self.max_level = int(min_level), int(max_level)
        assert len(scales) == self.max_level - self.min_level + 1
        self.canonical_level, self.canonical_box_size = canonical_level, canonical_box_size

    def forward(self, x, box_lists, grad_scale=1.0):
        """x: list of NHWC maps; box_lists: list[Boxes] per image -> [R, P, P, C] bf16."""
        dev = x[0].device
        if isinstance(box_lists, torch.Tensor):  # stacked [B, S, 4]
            nb, ns = box_lists.shape[:2]
            boxes = box_lists.reshape(-1, 4)
            idx = torch.arange(nb, device=dev, dtype=torch.float32)[:, None].expand(nb, ns).reshape(-1)
        else:
            sizes = [len(b) for b in box_lists]
            boxes = torch.cat([b.tensor for b in box_lists], dim=0)
            idx = image_index(sizes, dev)
        rois = torch.cat([idx[:, None], boxes], dim=1).contiguous()
        if len(self.scales) == 1:
            levels = torch.zeros(rois.shape[0], dtype=torch.int32, device=dev)
        else:
            levels = F.assign_levels(boxes, self.min_level, self.max_level, self.canonical_box_size, self.canonical_level)
        if rois.shape[0] == 0:
            return torch.zeros((0, self.output_size, self.output_size, x[0].shape[3]), dtype=x[0].dtype, device=dev)
        return F.roi_align(x, rois, levels, self.output_size, self.scales, grad_scale)


# ---------------------------------------------------------------------------------------------
@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Module):
    """flatten -> fc1 -> ReLU -> fc2 -> ReLU (box_head.py:26-118); fc1 runs as a PxP "conv" over the pooled
    NHWC tile so the reference's (c, h, w) flatten order is kept without a transpose."""

    @configurable
    def __init__(self, input_shape, *, conv_dims, fc_dims, conv_norm=""):
        super().__init__()
        assert len(conv_dims) == 0 and len(fc_dims) > 0 and conv_norm == ""
        self._in = (input_shape.channels, input_shape.height, input_shape.width)
        self.fcs = []
        dim = input_shape.channels * input_shape.height * input_shape.width
        for k, fc_dim in enumerate(fc_dims):
            fc = Linear(dim, fc_dim)
            self.add_module("fc{}".format(k + 1), fc)
            self.fcs.append(fc)
            dim = fc_dim
        self._out_dim = dim
        for layer in self.fcs:
            c2_xavier_fill(layer)

    @classmethod
    def from_config(cls, cfg, input_shape):
        num_conv, conv_dim = cfg.MODEL.ROI_BOX_HEAD.NUM_CONV, cfg.MODEL.ROI_BOX_HEAD.CONV_DIM
        num_fc, fc_dim = cfg.MODEL.ROI_BOX_HEAD.NUM_FC, cfg.MODEL.ROI_BOX_HEAD.FC_DIM
        return {"input_shape": input_shape, "conv_dims": [conv_dim] * num_conv, "fc_dims": [fc_dim] * num_fc,
                "conv_norm": cfg.MODEL.ROI_BOX_HEAD.NORM}

    def forward(self, x):
        c, h, w = self._in
        fc1 = self.fcs[0]
        y = F.conv2d(x, fc1.weight.view(fc1.out_features, c, h, w), fc1.bias, 1, 0, relu=True, param=fc1.weight)
        y = y.view(y.shape[0], -1)
        for fc in self.fcs[1:]:
            y = fc(y, relu=True)
        return y

    @property
    def output_shape(self):
        return ShapeSpec(channels=self._out_dim)


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)


class FastRCNNOutputLayers(nn.Module):
    """cls_score (K+1) and bbox_pred (4 when class agnostic) + losses / inference (fast_rcnn.py:174-569)."""

    @configurable
    def __init__(self, input_shape, *, box2box_weights, num_classes, test_score_thresh=0.0, test_nms_thresh=0.5,
                 test_topk_per_image=100, cls_agnostic_bbox_reg=False, smooth_l1_beta=0.0,
                 box_reg_loss_type="smooth_l1", loss_weight=1.0):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        self.num_classes = num_classes
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.cls_score = Linear(input_size, num_classes + 1)
        num_bbox_reg_classes = 1 if cls_agnostic_bbox_reg else num_classes
        assert cls_agnostic_bbox_reg, "the U2Seg configs use class-agnostic box regression"
        self.bbox_pred = Linear(input_size, num_bbox_reg_classes * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for layer in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(layer.bias, 0)
        self.box2box_weights = tuple(box2box_weights)
        assert box_reg_loss_type == "smooth_l1" and smooth_l1_beta == 0.0
        self.test_score_thresh, self.test_nms_thresh = test_score_thresh, test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        if isinstance(loss_weight, float):
            loss_weight = {"loss_cls": loss_weight, "loss_box_reg": loss_weight}
        self.loss_weight = loss_weight

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {
            "input_shape": input_shape,
            "box2box_weights": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS,
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "cls_agnostic_bbox_reg": cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
            "smooth_l1_beta": cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
            "test_score_thresh": cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            "test_nms_thresh": cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
            "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "box_reg_loss_type": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
            "loss_weight": {"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
        }

    def forward(self, x):
        """x [R, 1024] -> scores [R, 832] (K+1 valid), deltas [R, 32] (4 valid)."""
        return self.cls_score(x), self.bbox_pred(x)

    def losses(self, predictions, proposals):
        scores, deltas = predictions
        if isinstance(proposals, BatchList) and proposals.stacked and proposals.gt_boxes is not None:
            gt_classes = proposals.gt_classes.reshape(-1)
            proposal_boxes, gt_boxes = proposals.boxes.reshape(-1, 4), proposals.gt_boxes.reshape(-1, 4)
        else:
            gt_classes = torch.cat([p.gt_classes for p in proposals], dim=0) if len(proposals) else torch.empty(0)
            proposal_boxes = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
            gt_boxes = torch.cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], dim=0)
        r = gt_classes.numel()
        if r == 0:
            z = scores.float().sum() * 0.0
            return {"loss_cls": z, "loss_box_reg": deltas.float().sum() * 0.0}
        loss_cls = F.softmax_cross_entropy(scores, gt_classes, self.num_classes + 1)
        loss_box = F.box_reg_l1_loss(deltas, proposal_boxes, gt_boxes, gt_classes, self.num_classes,
                                     self.box2box_weights, max(r, 1.0))
        losses = {"loss_cls": loss_cls, "loss_box_reg": loss_box}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

    def predict_boxes(self, predictions, proposals, stacked=False):
        """stacked=True (BatchList input): one [B, S, 4] tensor instead of the per-image tuple."""
        _, deltas = predictions
        num_prop = [len(p) for p in proposals]
        if isinstance(proposals, BatchList) and proposals.stacked:
            proposal_boxes = proposals.boxes.reshape(-1, 4)
        else:
            assert not stacked or len(set(num_prop)) == 1
            proposal_boxes = torch.cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        if proposal_boxes.shape[0] == 0:
            return [proposal_boxes.new_zeros((0, 4)) for _ in proposals]
        boxes = F.apply_deltas(proposal_boxes, deltas[:, :4].float().contiguous(), self.box2box_weights, None, None,
                               _SCALE_CLAMP)
        if stacked:
            return boxes.view(len(num_prop), num_prop[0], 4)
        return boxes.split(num_prop)

    def predict_probs(self, predictions, proposals, split=True):
        """split=False: the [sum R_i, K+1] probabilities of the whole batch in one tensor."""
        scores, _ = predictions
        probs = torch.softmax(scores[:, : self.num_classes + 1].float(), dim=-1)
        if not split:
            return probs
        return probs.split([len(p) for p in proposals], dim=0)


# ---------------------------------------------------------------------------------------------
@ROI_MASK_HEAD_REGISTRY.register()
class MaskRCNNConvUpsampleHead(nn.Module):
    """4 x (conv3x3 + ReLU) -> deconv 2x2 s2 + ReLU -> 1x1 predictor to K channels (mask_head.py:215-290)."""

    @configurable
    def __init__(self, input_shape, *, num_classes, conv_dims, conv_norm="", loss_weight=1.0, vis_period=0):
        super().__init__()
        assert len(conv_dims) >= 1 and conv_norm == ""
        self.loss_weight, self.vis_period = loss_weight, vis_period
        self.conv_norm_relus = []
        cur = input_shape.channels
        for k, conv_dim in enumerate(conv_dims[:-1]):
            conv = Conv2d(cur, conv_dim, kernel_size=3, stride=1, padding=1, bias=True, activation="relu")
            self.add_module("mask_fcn{}".format(k + 1), conv)
            self.conv_norm_relus.append(conv)
            cur = conv_dim
        self.deconv = ConvTranspose2d(cur, conv_dims[-1], kernel_size=2, stride=2, padding=0)
        cur = conv_dims[-1]
        self.predictor = Conv2d(cur, num_classes, kernel_size=1, stride=1, padding=0)
        for layer in self.conv_norm_relus + [self.deconv]:
            c2_msra_fill(layer)
        nn.init.normal_(self.predictor.weight, std=0.001)
        nn.init.constant_(self.predictor.bias, 0)
        self.num_classes = num_classes

    @classmethod
    def from_config(cls, cfg, input_shape):
        conv_dim, num_conv = cfg.MODEL.ROI_MASK_HEAD.CONV_DIM, cfg.MODEL.ROI_MASK_HEAD.NUM_CONV
        ret = {"conv_dims": [conv_dim] * (num_conv + 1), "conv_norm": cfg.MODEL.ROI_MASK_HEAD.NORM, "input_shape": input_shape}
        ret["num_classes"] = 1 if cfg.MODEL.ROI_MASK_HEAD.CLS_AGNOSTIC_MASK else cfg.MODEL.ROI_HEADS.NUM_CLASSES
        return ret

    def trunk(self, x, shuffle=True):
        for layer in self.conv_norm_relus:
            x = layer(x)
        return self.deconv(x, relu=True, shuffle=shuffle)

    def forward(self, x, instances):
        """Training: {"loss_mask"}; inference: adds pred_masks [n,1,2P,2P] to the instances (mask_head.py:186-212)."""
        if self.training:
            # (the deconvolution's phases go to the loss kernel unshuffled, as at inference: no pixel-shuffle copy either way)
            return {"loss_mask": self.mask_loss(self.trunk(x, shuffle=False), instances, phased=True) * self.loss_weight}
        # inference: the deconvolution's phases go to the predictor unshuffled, and only the predicted class's channel is formed
        self.mask_inference(self.trunk(x, shuffle=False), instances, phased=True)
        return instances

    def mask_loss(self, x, instances, phased=False):
        """mask_rcnn_loss (mask_head.py:33-112) fused with the predictor: only the gt-class channel is formed.  x: the trunk
        output [n, 2P, 2P, C], or with `phased` the deconvolution's unshuffled phases [n, P, P, 4 C]."""
        side = x.shape[1] * 2 if phased else x.shape[1]
        from ..structures.masks import crop_and_resize_batch

        keep = [inst for inst in instances if len(inst) > 0]
        if len(keep) == 0:
            return x.float().sum() * 0.0 + self.predictor.weight.sum() * 0.0
        gt_classes = torch.cat([inst.gt_classes.to(torch.int64) for inst in keep], dim=0)
        gt_masks = crop_and_resize_batch([inst.gt_masks for inst in keep], [inst.proposal_boxes.tensor for inst in keep],
                                         side).to(torch.uint8)
        if self.num_classes == 1:
            gt_classes = torch.zeros_like(gt_classes)
        return F.mask_predict_bce_loss(x, self.predictor.weight, self.predictor.bias, gt_classes, gt_masks, phased)

    def mask_inference(self, x, pred_instances, phased=False):
        """mask_rcnn_inference (mask_head.py:115-158): sigmoid of the predicted-class channel.  x: the trunk output
        [n, 2P, 2P, C], or with `phased` the deconvolution's unshuffled phases [n, P, P, 4 C]."""
        n = x.shape[0]
        if self.num_classes == 1 or n == 0:
            cls = torch.zeros(n, dtype=torch.long, device=x.device)
        else:
            cls = torch.cat([i.pred_classes for i in pred_instances])
        probs = F.mask_predict_prob(x, self.predictor.weight, self.predictor.bias, cls, phased=phased)
        for prob, inst in zip(probs.split([len(i) for i in pred_instances], dim=0), pred_instances):
            inst.pred_masks = prob


def build_mask_head(cfg, input_shape):
    return ROI_MASK_HEAD_REGISTRY.get(cfg.MODEL.ROI_MASK_HEAD.NAME)(cfg, input_shape)


# ---------------------------------------------------------------------------------------------
def select_foreground_proposals(proposals, bg_label):
    """roi_heads.py:46-75 (one device->host sync for the whole batch instead of one per image)."""
    if isinstance(proposals, BatchList) and proposals.stacked and len(proposals):
        # stacked: compact the foreground rows of every image to the front with one stable sort, one count transfer
        gc = proposals.gt_classes
        fgm = (gc != -1) & (gc != bg_label)
        order = torch.argsort((~fgm).to(torch.int8), dim=1, stable=True)
        counts = fgm.sum(dim=1).tolist()
        # the stacked columns are reordered for the whole batch at once and the per-image tables take views of them (indexing every
        # column of every image was 80 launches per 16-image step); a column without a stacked form is indexed per image
        o4 = order[..., None].expand(-1, -1, 4)
        fast = {"proposal_boxes": torch.gather(proposals.boxes, 1, o4), "gt_classes": torch.gather(gc, 1, order)}
        if proposals.gt_boxes is not None:
            fast["gt_boxes"] = torch.gather(proposals.gt_boxes, 1, o4)
        if proposals.logits is not None:
            fast["objectness_logits"] = torch.gather(proposals.logits, 1, order)
        g_match = torch.gather(proposals.match, 1, order) if proposals.match is not None else None
        out = []
        for i, (p, c) in enumerate(zip(proposals, counts)):
            res = Instances(p.image_size)
            rows = None
            for name, col in p.get_fields().items():
                if name in fast and len(col) == order.shape[1]:
                    v = fast[name][i, :c]
                    res.set(name, Boxes(v) if isinstance(col, Boxes) else v)
                elif g_match is not None and isinstance(col, BitMasks) and col._index is not None \
                        and col._index.data_ptr() == proposals.match[i].data_ptr() and col._index.numel() == order.shape[1]:
                    res.set(name, BitMasks(col._base, g_match[i, :c]))  # the sampler's lazy row index, composed with the order
                else:
                    rows = order[i, :c] if rows is None else rows
                    res.set(name, col[rows])
            out.append(res)
        return out, list(fgm)
    masks = [(p.gt_classes != -1) & (p.gt_classes != bg_label) for p in proposals]
    counts = torch.stack([m.sum() for m in masks]).tolist() if masks else []
    idx_all = torch.nonzero(torch.cat(masks), as_tuple=True)[0] if masks else None
    fg, off, start = [], 0, 0
    for p, m, c in zip(proposals, masks, counts):
        fg.append(p[idx_all[start : start + c] - off])
        off += len(p)
        start += c
    return fg, masks


def add_ground_truth_to_proposals(targets, proposals):
    """proposal_utils.py:138-205: gt boxes appended after the proposals with logit log((1-1e-10)/1e-10)."""
    gt_logit_value = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
    out = []
    for gt, prop in zip(targets, proposals):
        res = Instances(prop.image_size)
        res.proposal_boxes = Boxes.cat([prop.proposal_boxes, gt.gt_boxes])
        gt_logits = gt_logit_value * torch.ones(len(gt), device=prop.objectness_logits.device)
        res.objectness_logits = torch.cat([prop.objectness_logits, gt_logits.to(prop.objectness_logits.dtype)])
        out.append(res)
    return out


class ROIHeads(nn.Module):
    @configurable
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_iou_threshold,
                 proposal_append_gt=True):
        super().__init__()
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction
        self.num_classes = num_classes
        self.proposal_iou_threshold = proposal_iou_threshold
        self.proposal_append_gt = proposal_append_gt

    @classmethod
    def from_config(cls, cfg):
        assert list(cfg.MODEL.ROI_HEADS.IOU_LABELS) == [0, 1] and len(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS) == 1
        return {
            "batch_size_per_image": cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "proposal_append_gt": cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT,
            "proposal_iou_threshold": cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS[0],
        }

    @staticmethod
    def _match(boxes, targets, thr):
        """IoU-match one image's boxes to its gt: (matched gt idx int64, labels {0,1} int8)."""
        n = boxes.shape[0]
        dev = boxes.device
        if len(targets) == 0 or n == 0:
            return torch.zeros(n, dtype=torch.int64, device=dev), torch.zeros(n, dtype=torch.int8, device=dev)
        gt = targets.gt_boxes.tensor[None].contiguous()
        ngt = device_constant([len(targets)], torch.int32, dev)
        match, labels, _ = F.iou_match(boxes[None].contiguous(), gt, ngt, thr, thr, False)
        return match[0].long(), labels[0]

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets):
        """roi_heads.py:220-302.  Without an injected permutation source (parity tests) the whole batch is labelled and
        sampled on padded tensors: one IoU-match launch, two top-k, one host synchronisation (the sample counts)."""
        from . import sampling

        if sampling.permutation_source() is None and len(proposals) and all(t.has("gt_classes") for t in targets):
            return self._label_and_sample_padded(proposals_from_list(proposals, self.training), targets)
        if self.proposal_append_gt:
            proposals = add_ground_truth_to_proposals(targets, proposals)
        out = []
        for prop, tgt in zip(proposals, targets):
            has_gt = len(tgt) > 0
            matched_idxs, matched_labels = self._match(prop.proposal_boxes.tensor, tgt, self.proposal_iou_threshold)
            if has_gt:
                gt_classes = tgt.gt_classes[matched_idxs]
                gt_classes[matched_labels == 0] = self.num_classes
                gt_classes[matched_labels == -1] = -1
            else:
                gt_classes = torch.zeros_like(matched_idxs) + self.num_classes
            fg_idx, bg_idx = subsample_labels(gt_classes, self.batch_size_per_image, self.positive_fraction, self.num_classes)
            sampled = torch.cat([fg_idx, bg_idx], dim=0)
            res = prop[sampled]
            res.gt_classes = gt_classes[sampled]
            if has_gt:
                sampled_targets = matched_idxs[sampled]
                for name, value in tgt.get_fields().items():
                    if name.startswith("gt_") and not res.has(name):
                        res.set(name, value[sampled_targets])
            out.append(res)
        return out

    def _label_and_sample_padded(self, lp, targets):
        dev = lp.boxes.device
        nb, npad = lp.boxes.shape[:2]
        pt = PaddedTargets.of(targets, dev)
        g = pt.boxes.shape[1]
        if self.proposal_append_gt:  # proposal_utils.py:138-205; the gt rows follow the (padded) proposal rows
            gt_logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
            boxes = torch.cat([lp.boxes, pt.boxes], dim=1)
            logits = torch.cat([lp.logits, lp.logits.new_full((nb, g), gt_logit)], dim=1)
            ar = torch.arange(npad + g, device=dev)[None]
            valid = (ar < lp.counts[:, None]) | ((ar >= npad) & (ar < npad + pt.counts[:, None]))
        else:
            boxes, logits = lp.boxes, lp.logits
            valid = torch.arange(npad, device=dev)[None] < lp.counts[:, None]
        n = boxes.shape[1]
        thr = self.proposal_iou_threshold
        match, labels, _ = F.iou_match(boxes, pt.boxes, pt.counts, thr, thr, False)
        match = match.long()
        gt_classes = torch.gather(pt.classes, 1, match)
        gt_classes = torch.where(labels == 1, gt_classes, torch.where(labels == 0, self.num_classes, -1))
        gt_classes = torch.where(valid, gt_classes, -1)  # padding rows are never sampled
        # sampling.py:38-54 as random keys + top-k: <= 128 random foreground rows, background fills up to 512
        total = self.batch_size_per_image
        max_fg = int(total * self.positive_fraction)
        from . import sampling

        key = sampling.random_keys((nb, n), dev)
        is_bg = gt_classes == self.num_classes
        is_fg = (gt_classes >= 0) & ~is_bg
        kind = (is_fg.to(torch.int8) + is_bg.to(torch.int8) * 2).contiguous()  # 1 foreground, 2 background, 0 not sampled
        kf, kb = min(max_fg, n), min(total, n)
        # the smallest keys first, ties by row index: positive[argsort(key[positive])] of sampling.py:38-54
        (_, fg_idx, fg_cnt), (_, bg_idx, bg_cnt) = F.topk_rows_multi([
            dict(vals=key, k=kf, largest=False, mask=kind, mask_value=1, want_vals=False),
            dict(vals=key, k=kb, largest=False, mask=kind, mask_value=2, want_vals=False)])
        num_bg = torch.minimum(bg_cnt, total - fg_cnt)
        fg_valid = torch.arange(kf, device=dev)[None] < fg_cnt[:, None]
        bg_valid = torch.arange(kb, device=dev)[None] < num_bg[:, None]
        cand, cvalid = torch.cat([fg_idx, bg_idx], dim=1).long(), torch.cat([fg_valid, bg_valid], dim=1)
        order = torch.argsort((~cvalid).to(torch.int8), dim=1, stable=True)[:, :total]
        sampled = torch.gather(cand, 1, order)  # [B, S]; rows >= count are padding
        s = sampled.shape[1]
        flags = [cvalid.sum(dim=1)] + ([lp.finite.reshape(1).to(torch.int64)] if lp.finite is not None else [])
        vals = torch.cat(flags).tolist()  # the one host synchronisation of the sampler
        F.issue_deferred_piece()  # the chip is idle from here until the first stage's kernels are launched (functional.defer_pieces)
        if lp.finite is not None:
            check_finite(bool(vals[-1]), self.training)
            lp.finite = None
        counts = vals[:nb]
        s_boxes = torch.gather(boxes, 1, sampled[..., None].expand(-1, -1, 4))
        s_logits = torch.gather(logits, 1, sampled)
        s_cls = torch.gather(gt_classes, 1, sampled)
        s_match = torch.gather(match, 1, sampled)
        s_gtb = torch.gather(pt.boxes, 1, s_match[..., None].expand(-1, -1, 4))
        out = BatchList()
        for i, (size, tgt, c) in enumerate(zip(lp.image_sizes, targets, counts)):
            res = Instances(size)
            res.proposal_boxes = Boxes(s_boxes[i, :c])
            res.objectness_logits = s_logits[i, :c]
            res.gt_classes = s_cls[i, :c]
            if len(tgt) > 0:
                res.gt_boxes = Boxes(s_gtb[i, :c])
                for name, value in tgt.get_fields().items():
                    if name.startswith("gt_") and not res.has(name):
                        res.set(name, value[s_match[i, :c]])
            out.append(res)
        if all(c == s for c in counts):
            out.boxes, out.gt_classes, out.gt_boxes = s_boxes, s_cls, s_gtb
            out.logits, out.match = s_logits, s_match
        return out


class StandardROIHeads(ROIHeads):
    """roi_heads.py:530-846 (box + mask branches; keypoints are not used by U2Seg)."""

    @configurable
    def __init__(self, *, box_in_features, box_pooler, box_head, box_predictor, mask_in_features=None, mask_pooler=None,
                 mask_head=None, train_on_pred_boxes=False, **kwargs):
        super().__init__(**kwargs)
        self.in_features = self.box_in_features = box_in_features
        self.box_pooler, self.box_head, self.box_predictor = box_pooler, box_head, box_predictor
        self.mask_on = mask_in_features is not None
        if self.mask_on:
            self.mask_in_features, self.mask_pooler, self.mask_head = mask_in_features, mask_pooler, mask_head
        assert not train_on_pred_boxes

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = super().from_config(cfg)
        ret["train_on_pred_boxes"] = cfg.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES
        ret.update(cls._init_box_head(cfg, input_shape))
        if cfg.MODEL.MASK_ON:
            ret.update(cls._init_mask_head(cfg, input_shape))
        return ret

    @classmethod
    def _init_box_head(cls, cfg, input_shape):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        res = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        in_channels = [input_shape[f].channels for f in in_features][0]
        box_pooler = ROIPooler(res, scales, cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO, cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE)
        box_head = build_box_head(cfg, ShapeSpec(channels=in_channels, height=res, width=res))
        box_predictor = FastRCNNOutputLayers(cfg, box_head.output_shape)
        return {"box_in_features": in_features, "box_pooler": box_pooler, "box_head": box_head, "box_predictor": box_predictor}

    @classmethod
    def _init_mask_head(cls, cfg, input_shape):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        res = cfg.MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION
        scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        in_channels = [input_shape[f].channels for f in in_features][0]
        ret = {"mask_in_features": in_features}
        ret["mask_pooler"] = ROIPooler(res, scales, cfg.MODEL.ROI_MASK_HEAD.POOLER_SAMPLING_RATIO, cfg.MODEL.ROI_MASK_HEAD.POOLER_TYPE)
        ret["mask_head"] = build_mask_head(cfg, ShapeSpec(channels=in_channels, width=res, height=res))
        return ret

    def forward(self, images, features, proposals, targets=None):
        if self.training:
            assert targets, "'targets' argument is required during training"
            proposals = self.label_and_sample_proposals(proposals, targets)
            features = self._tap_pooled_features(features)
            losses = self._forward_box(features, proposals)
            losses.update(self._forward_mask(features, proposals))
            return proposals, losses
        pred_instances = self._forward_box(features, proposals)
        pred_instances = self.forward_with_given_boxes(features, pred_instances)
        return pred_instances, {}

    def forward_with_given_boxes(self, features, instances):
        assert not self.training
        assert instances[0].has("pred_boxes") and instances[0].has("pred_classes")
        return self._forward_mask(features, instances)

    def _forward_box(self, features, proposals):
        feats = [features[f] for f in self.box_in_features]
        box_features = self.box_head(self.box_pooler(feats, [x.proposal_boxes for x in proposals]))
        predictions = self.box_predictor(box_features)
        if self.training:
            return self.box_predictor.losses(predictions, proposals)
        from .inference import fast_rcnn_inference

        boxes = self.box_predictor.predict_boxes(predictions, proposals)
        scores = self.box_predictor.predict_probs(predictions, proposals)
        pred, _ = fast_rcnn_inference(boxes, scores, [x.image_size for x in proposals], self.box_predictor.test_score_thresh,
                                      self.box_predictor.test_nms_thresh, self.box_predictor.test_topk_per_image)
        return pred

    def _tap_pooled_features(self, features):
        """All poolers of these heads read the same FPN maps: defer their ROIAlign backward passes to one gather."""
        names = list(self.box_in_features)
        if torch.is_grad_enabled() and (not self.mask_on or list(self.mask_in_features) == names):
            tapped = F.roi_grad_tap([features[f] for f in names])
            features = dict(features)
            features.update(zip(names, tapped))
        return features

    def _forward_mask(self, features, instances):
        if not self.mask_on:
            return {} if self.training else instances
        if self.training:
            instances, _ = select_foreground_proposals(instances, self.num_classes)
        feats = [features[f] for f in self.mask_in_features]
        boxes = [x.proposal_boxes if self.training else x.pred_boxes for x in instances]
        pooled = self.mask_pooler(feats, boxes)
        return self.mask_head(pooled, instances)


@ROI_HEADS_REGISTRY.register()
class CascadeROIHeads(StandardROIHeads):
    """cascade_rcnn.py:32-299: three box stages with rising IoU thresholds, gradients into the shared features
    scaled by 1/3 (the _ScaleGradient of cascade_rcnn.py:20-28 is folded into the ROIAlign backward)."""

    @configurable
    def __init__(self, *, box_in_features, box_pooler, box_heads, box_predictors, proposal_iou_thresholds, **kwargs):
        assert "proposal_iou_threshold" not in kwargs or True
        num_stages = self.num_cascade_stages = len(box_heads)
        box_heads = nn.ModuleList(box_heads)
        box_predictors = nn.ModuleList(box_predictors)
        assert len(box_predictors) == num_stages and len(proposal_iou_thresholds) == num_stages
        super().__init__(box_in_features=box_in_features, box_pooler=box_pooler, box_head=box_heads,
                         box_predictor=box_predictors, **kwargs)
        self.cascade_ious = tuple(proposal_iou_thresholds)

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = super().from_config(cfg, input_shape)
        ret.pop("train_on_pred_boxes", None)
        return ret

    @classmethod
    def _init_box_head(cls, cfg, input_shape):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        res = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        cascade_bbox_reg_weights = cfg.MODEL.ROI_BOX_CASCADE_HEAD.BBOX_REG_WEIGHTS
        cascade_ious = cfg.MODEL.ROI_BOX_CASCADE_HEAD.IOUS
        assert len(cascade_bbox_reg_weights) == len(cascade_ious)
        assert cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG, "CascadeROIHeads only support class-agnostic regression now!"
        assert cascade_ious[0] == cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS[0]
        in_channels = [input_shape[f].channels for f in in_features][0]
        box_pooler = ROIPooler(res, scales, cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO, cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE)
        pooled_shape = ShapeSpec(channels=in_channels, width=res, height=res)
        box_heads, box_predictors = [], []
        for bbox_reg_weights in cascade_bbox_reg_weights:
            box_head = build_box_head(cfg, pooled_shape)
            box_heads.append(box_head)
            box_predictors.append(FastRCNNOutputLayers(cfg, box_head.output_shape, box2box_weights=tuple(bbox_reg_weights)))
        return {"box_in_features": in_features, "box_pooler": box_pooler, "box_heads": box_heads,
                "box_predictors": box_predictors, "proposal_iou_thresholds": cascade_ious}

    def forward(self, images, features, proposals, targets=None):
        if self.training:
            proposals = self.label_and_sample_proposals(proposals, targets)
            features = self._tap_pooled_features(features)
            # the mask head only needs the sampled proposals: it runs on its own stream beside the three cascade stages, whose
            # decode / match / relabel steps between the stages leave most of the chip idle
            stacked = self.mask_on and isinstance(proposals, BatchList) and proposals.stacked and proposals.boxes.is_cuda
            aux = F.aux_stream(proposals.boxes.device, 1) if stacked and os.environ.get("U2_MASK_STREAM", "1") != "0" else None
            if aux is not None:
                main = torch.cuda.current_stream(proposals.boxes.device)
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    mask_losses = self._forward_mask(features, proposals)
                losses = self._forward_box(features, proposals, targets)
                main.wait_stream(aux)
                losses.update(mask_losses)
            else:
                losses = self._forward_box(features, proposals, targets)
                losses.update(self._forward_mask(features, proposals))
            return proposals, losses
        pred_instances = self._forward_box(features, proposals)
        pred_instances = self.forward_with_given_boxes(features, pred_instances)
        return pred_instances, {}

    def _forward_box(self, features, proposals, targets=None):
        feats = [features[f] for f in self.box_in_features]
        head_outputs = []
        prev_pred_boxes = None
        image_sizes = [x.image_size for x in proposals]
        for k in range(self.num_cascade_stages):
            if k > 0:
                if isinstance(prev_pred_boxes, torch.Tensor):  # stacked [B, S, 4] (training, equal counts per image)
                    proposals = self._next_stage_stacked(prev_pred_boxes, image_sizes, k, targets)
                else:
                    proposals = self._create_proposals_from_boxes(prev_pred_boxes, image_sizes)
                    if self.training:
                        proposals = self._match_and_label_boxes(proposals, k, targets)
            predictions = self._run_stage(feats, proposals, k)
            stacked = self.training and isinstance(proposals, BatchList) and proposals.stacked
            prev_pred_boxes = self.box_predictor[k].predict_boxes(predictions, proposals, stacked=stacked)
            head_outputs.append((self.box_predictor[k], predictions, proposals))
        if self.training:
            losses = {}
            F.issue_deferred_piece()
            for stage, (predictor, predictions, props) in enumerate(head_outputs):
                stage_losses = predictor.losses(predictions, props)
                losses.update({k + "_stage{}".format(stage): v for k, v in stage_losses.items()})
            return losses
        from .inference import fast_rcnn_inference

        # cascade_rcnn.py:155-161: per image sum(scores of the stages) * (1 / stages).  The sum of the whole batch at once (the
        # same additions in the same order, 0 + s0 being s0): per image it was four launches on a [1000, K+1] matrix, 128 per
        # 32-image batch; with equal proposal counts the stacked [B, R, .] tensors go to the filter without a copy
        avg = None
        for predictor, predictions, props in head_outputs:
            p = predictor.predict_probs(predictions, props, split=False)
            avg = p if avg is None else avg + p
        avg = avg * (1.0 / self.num_cascade_stages)
        predictor, predictions, proposals = head_outputs[-1]
        num_inst = [len(p) for p in proposals]
        if len(set(num_inst)) == 1 and num_inst[0] > 0:
            scores = avg.view(len(num_inst), num_inst[0], avg.shape[1])
            boxes = predictor.predict_boxes(predictions, proposals, stacked=True)
        else:
            scores = avg.split(num_inst, dim=0)
            boxes = predictor.predict_boxes(predictions, proposals)
        pred, _ = fast_rcnn_inference(boxes, scores, image_sizes, predictor.test_score_thresh, predictor.test_nms_thresh,
                                      predictor.test_topk_per_image)
        return pred

    @torch.no_grad()
    def _match_and_label_boxes(self, proposals, stage, targets):
        for prop, tgt in zip(proposals, targets):
            matched_idxs, labels = self._match(prop.proposal_boxes.tensor, tgt, self.cascade_ious[stage])
            if len(tgt) > 0:
                gt_classes = tgt.gt_classes[matched_idxs]
                gt_classes[labels == 0] = self.num_classes
                gt_boxes = tgt.gt_boxes[matched_idxs]
            else:
                gt_classes = torch.zeros_like(matched_idxs) + self.num_classes
                gt_boxes = Boxes(tgt.gt_boxes.tensor.new_zeros((len(prop), 4)))
            prop.gt_classes = gt_classes
            prop.gt_boxes = gt_boxes
        return proposals

    @torch.no_grad()
    def _next_stage_stacked(self, boxes, image_sizes, stage, targets):
        """cascade_rcnn.py:226-299 for a batch whose images all carry S boxes: clip, (rarely) drop empty boxes, IoU-match
        and label with a handful of batched launches and one host synchronisation (the any-empty flag)."""
        nb, ns = boxes.shape[:2]
        dev = boxes.device
        lim = device_constant([[[w, h, w, h]] for h, w in image_sizes], torch.float32, dev)
        boxes = torch.minimum(boxes.detach().clamp(min=0), lim)
        nonempty = (boxes[..., 2:] > boxes[..., :2]).all(dim=-1)
        if not bool(nonempty.all()):  # cascade_rcnn.py:291-294: ragged result, take the per-image path
            props = self._create_proposals_from_boxes(list(boxes), image_sizes)
            return self._match_and_label_boxes(props, stage, targets)
        F.issue_deferred_piece()  # behind the host synchronisation: match / relabel below are ~20 tiny launches
        pt = PaddedTargets.of(targets, dev)
        thr = self.cascade_ious[stage]
        match, labels, _ = F.iou_match(boxes.contiguous(), pt.boxes, pt.counts, thr, thr, False)
        match = match.long()
        gt_classes = torch.where(labels == 1, torch.gather(pt.classes, 1, match), self.num_classes)
        gt_boxes = torch.gather(pt.boxes, 1, match[..., None].expand(-1, -1, 4))
        out = BatchList()
        for i, size in enumerate(image_sizes):
            prop = Instances(size)
            prop.proposal_boxes = Boxes(boxes[i])
            prop.gt_classes = gt_classes[i]
            prop.gt_boxes = Boxes(gt_boxes[i])
            out.append(prop)
        out.boxes, out.gt_classes, out.gt_boxes = boxes, gt_classes, gt_boxes
        return out

    def _run_stage(self, feats, proposals, stage):
        gs = 1.0 / self.num_cascade_stages if self.training else 1.0
        if isinstance(proposals, BatchList) and proposals.stacked:
            box_features = self.box_pooler(feats, proposals.boxes, grad_scale=gs)
            return self.box_predictor[stage](self.box_head[stage](box_features))
        box_features = self.box_pooler(feats, [x.proposal_boxes for x in proposals], grad_scale=gs)
        return self.box_predictor[stage](self.box_head[stage](box_features))

    def _create_proposals_from_boxes(self, boxes, image_sizes):
        out = []
        # Boxes.clip for all images at once (five small launches per image otherwise: 320 per 32-image batch and cascade step)
        counts = [int(b.shape[0]) for b in boxes]
        if sum(counts):
            dev = boxes[0].device
            lim = device_constant([[s[1], s[0], s[1], s[0]] for s in image_sizes], torch.float32, dev)
            rows = torch.repeat_interleave(device_constant(list(range(len(counts))), torch.int64, dev),
                                           device_upload(counts, torch.int64, dev), output_size=sum(counts))
            flat = torch.minimum(torch.cat([b.detach() for b in boxes]).float().clamp(min=0), lim[rows])
            clipped = [Boxes(t) for t in flat.split(counts)]
        else:
            clipped = [Boxes(b.detach()) for b in boxes]
        if self.training:
            # drop empty boxes (cascade_rcnn.py:291-294); a single sync decides whether any image needs the filter
            keeps = [bx.nonempty() for bx in clipped]
            if not bool(torch.stack([k.all() for k in keeps]).all()):
                clipped = [bx[k] for bx, k in zip(clipped, keeps)]
        for bx, image_size in zip(clipped, image_sizes):
            prop = Instances(image_size)
            prop.proposal_boxes = bx
            out.append(prop)
        return out


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)


ROI_HEADS_REGISTRY.register(StandardROIHeads)
