"""Region proposal network: anchors, head, anchor labelling, losses, proposal decoding/NMS
(detectron2/modeling/anchor_generator.py:39-231, proposal_generator/rpn.py:67-533,
proposal_generator/proposal_utils.py:22-205).

Device work goes through the HIP kernels: shared 3x3 conv + 1x1 heads (conv_igemm), anchor<->gt IoU matching
(u2_iou_match), the fused per-level loss (u2_rpn_loss_level), box decoding (u2_apply_deltas) and per-level NMS
(u2_batched_nms), per-level top-k, score sort and anchor subsampling (u2_topk_rows).  torch is left with gathers / concatenation."""
import math

import torch
from torch import nn

from ..config import configurable
from ..layers import Conv2d
from ..layers import functional as F
from ..structures import Boxes, Instances
from ..utils.registry import Registry
from .batched import LazyProposals, PaddedTargets, device_constant
from .sampling import subsample_labels

ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")

_SCALE_CLAMP = math.log(1000.0 / 16)


def _broadcast_params(params, num_features, name):
    assert isinstance(params, (list, tuple)) and len(params)
    if not isinstance(params[0], (list, tuple)):
        return [params] * num_features
    if len(params) == 1:
        return list(params) * num_features
    assert len(params) == num_features, "{} has {} entries for {} feature maps".format(name, len(params), num_features)
    return params


@ANCHOR_GENERATOR_REGISTRY.register()
class DefaultAnchorGenerator(nn.Module):
    box_dim = 4

    @configurable
    def __init__(self, *, sizes, aspect_ratios, strides, offset=0.5):
        super().__init__()
        self.strides = strides
        self.num_features = len(strides)
        sizes = _broadcast_params(sizes, self.num_features, "sizes")
        aspect_ratios = _broadcast_params(aspect_ratios, self.num_features, "aspect_ratios")
        cells = [self.generate_cell_anchors(s, a).float() for s, a in zip(sizes, aspect_ratios)]
        for i, c in enumerate(cells):  # non-persistent buffers named like detectron2's BufferList
            self.register_buffer("cell_anchors_{}".format(i), c, persistent=False)
        self.offset = offset
        assert 0.0 <= self.offset < 1.0
        self._cache = {}

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {
            "sizes": cfg.MODEL.ANCHOR_GENERATOR.SIZES,
            "aspect_ratios": cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
            "strides": [x.stride for x in input_shape],
            "offset": cfg.MODEL.ANCHOR_GENERATOR.OFFSET,
        }

    @property
    def cell_anchors(self):
        return [getattr(self, "cell_anchors_{}".format(i)) for i in range(self.num_features)]

    @property
    def num_anchors(self):
        return [len(c) for c in self.cell_anchors]

    @staticmethod
    def generate_cell_anchors(sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
        anchors = []
        for size in sizes:
            area = size ** 2.0
            for aspect_ratio in aspect_ratios:
                w = math.sqrt(area / aspect_ratio)
                h = aspect_ratio * w
                anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(anchors)

    def grid_anchors(self, grid_sizes):
        out = []
        for size, stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            key = (tuple(size), stride, base.device)
            if key not in self._cache:
                gh, gw = size
                shifts_x = torch.arange(self.offset * stride, gw * stride, step=stride, dtype=torch.float32, device=base.device)
                shifts_y = torch.arange(self.offset * stride, gh * stride, step=stride, dtype=torch.float32, device=base.device)
                shift_y, shift_x = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
                shift_x, shift_y = shift_x.reshape(-1), shift_y.reshape(-1)
                shifts = torch.stack((shift_x, shift_y, shift_x, shift_y), dim=1)
                self._cache[key] = (shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4).contiguous()
            out.append(self._cache[key])
        return out

    def forward(self, grid_sizes):
        """grid_sizes: list of (H_i, W_i) -> list[Boxes] (H_i*W_i*A x 4)."""
        return [Boxes(x) for x in self.grid_anchors(grid_sizes)]


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    """3x3 conv + ReLU, then 1x1 objectness (A) and 1x1 deltas (4A), shared by all levels (rpn.py:67-177)."""

    @configurable
    def __init__(self, *, in_channels, num_anchors, box_dim=4, conv_dims=(-1,)):
        super().__init__()
        assert len(conv_dims) == 1
        out_channels = in_channels if conv_dims[0] == -1 else conv_dims[0]
        self.conv = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, activation="relu")
        self.objectness_logits = Conv2d(out_channels, num_anchors, kernel_size=1, stride=1)
        self.anchor_deltas = Conv2d(out_channels, num_anchors * box_dim, kernel_size=1, stride=1)
        for layer in [self.conv, self.objectness_logits, self.anchor_deltas]:
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)
        self.num_anchors = num_anchors
        self.fuse_predictors = True  # both 1x1 predictors as one conv (forward()); False: two convs, as the reference runs them

    @classmethod
    def from_config(cls, cfg, input_shape):
        in_channels = [s.channels for s in input_shape]
        assert len(set(in_channels)) == 1, "Each level must have the same channel!"
        anchor_generator = build_anchor_generator(cfg, input_shape)
        num_anchors = anchor_generator.num_anchors
        assert len(set(num_anchors)) == 1
        return {"in_channels": in_channels[0], "num_anchors": num_anchors[0], "box_dim": anchor_generator.box_dim,
                "conv_dims": cfg.MODEL.RPN.CONV_DIMS}

    def forward(self, features):
        """features: list of NHWC maps -> (list [B,H,W,32] objectness (A valid), list [B,H,W,>=4A] deltas (4A valid)).

        With A = 3 the two 1x1 predictors run as ONE conv of 15 output channels (rpn.py:170-176 reads `t` twice): the map holds
        the objectness in columns 0-2 and the deltas in columns 3-14, the second list holds views of it.  One read of `t`
        instead of two, and in training one data-gradient conv instead of two plus the sum autograd forms for a tensor with
        two consumers (550 MB at the stride-4 level); the loss kernel writes both gradients into one map (F.rpn_losses)."""
        objs, dlts = [], []
        a = self.num_anchors
        fused = self.fuse_predictors and a == 3 and self.anchor_deltas.out_channels == 4 * a
        if fused:
            w, bias = self._fused_predictor()
        for x in features:
            t = self.conv(x)
            if fused:
                y = F.conv2d(t, w, bias, 1, 0, param=w)
                y._u2_rpn_fused = True
                objs.append(y)
                dlts.append(y[..., a:])
            else:
                objs.append(self.objectness_logits(t))
                dlts.append(self.anchor_deltas(t))
        return objs, dlts

    def _fused_predictor(self):
        """[A + 4A, C, 1, 1] weight and bias of the two predictors.  Training: a fresh concatenation per forward pass (autograd
        splits its gradient), kernel layouts cached on the tensor for the five levels.  Inference: kept until a weight changes."""
        wo, wd = self.objectness_logits.weight, self.anchor_deltas.weight
        bo, bd = self.objectness_logits.bias, self.anchor_deltas.bias
        if torch.is_grad_enabled() and (wo.requires_grad or wd.requires_grad):
            w, b = torch.cat([wo, wd], 0), torch.cat([bo, bd], 0)
            w._u2_step_layouts = {}
            b._u2_step_bias = {}
            return w, b
        stamp = getattr(wo, "_u2_stamp", None)
        key = (wo._version, wd._version, bo._version, bd._version, wo.data_ptr(), wd.data_ptr(), stamp[0] if stamp else None)
        cached = self.__dict__.get("_fused_eval")
        if cached is None or cached[0] != key:
            w, b = torch.cat([wo.detach(), wd.detach()], 0), torch.cat([bo.detach(), bd.detach()], 0)
            w._u2_step_layouts = {}
            b._u2_step_bias = {}
            cached = self.__dict__["_fused_eval"] = (key, w, b)
        return cached[1], cached[2]


def build_rpn_head(cfg, input_shape):
    return RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, input_shape)


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPN(nn.Module):
    @configurable
    def __init__(self, *, in_features, head, anchor_generator, iou_thresholds, iou_labels, batch_size_per_image,
                 positive_fraction, pre_nms_topk, post_nms_topk, nms_thresh=0.7, min_box_size=0.0,
                 anchor_boundary_thresh=-1.0, loss_weight=1.0, box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0,
                 bbox_reg_weights=(1.0, 1.0, 1.0, 1.0)):
        super().__init__()
        self.in_features = in_features
        self.rpn_head = head
        self.anchor_generator = anchor_generator
        assert list(iou_labels) == [0, -1, 1] and len(iou_thresholds) == 2
        self.iou_thresholds = tuple(iou_thresholds)
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.pre_nms_topk = {True: pre_nms_topk[0], False: pre_nms_topk[1]}
        self.post_nms_topk = {True: post_nms_topk[0], False: post_nms_topk[1]}
        self.nms_thresh = nms_thresh
        self.min_box_size = float(min_box_size)
        self.anchor_boundary_thresh = anchor_boundary_thresh
        if isinstance(loss_weight, float):
            loss_weight = {"loss_rpn_cls": loss_weight, "loss_rpn_loc": loss_weight}
        self.loss_weight = loss_weight
        assert box_reg_loss_type == "smooth_l1" and smooth_l1_beta == 0.0, "the U2Seg configs use L1 (beta 0)"
        assert tuple(bbox_reg_weights) == (1.0, 1.0, 1.0, 1.0)
        self.bbox_reg_weights = tuple(bbox_reg_weights)

    @classmethod
    def from_config(cls, cfg, input_shape):
        in_features = cfg.MODEL.RPN.IN_FEATURES
        shapes = [input_shape[f] for f in in_features]
        return {
            "in_features": in_features,
            "min_box_size": cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE,
            "nms_thresh": cfg.MODEL.RPN.NMS_THRESH,
            "batch_size_per_image": cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.RPN.POSITIVE_FRACTION,
            "loss_weight": {"loss_rpn_cls": cfg.MODEL.RPN.LOSS_WEIGHT,
                            "loss_rpn_loc": cfg.MODEL.RPN.BBOX_REG_LOSS_WEIGHT * cfg.MODEL.RPN.LOSS_WEIGHT},
            "anchor_boundary_thresh": cfg.MODEL.RPN.BOUNDARY_THRESH,
            "bbox_reg_weights": cfg.MODEL.RPN.BBOX_REG_WEIGHTS,
            "box_reg_loss_type": cfg.MODEL.RPN.BBOX_REG_LOSS_TYPE,
            "smooth_l1_beta": cfg.MODEL.RPN.SMOOTH_L1_BETA,
            "pre_nms_topk": (cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.PRE_NMS_TOPK_TEST),
            "post_nms_topk": (cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TEST),
            "anchor_generator": build_anchor_generator(cfg, shapes),
            "iou_thresholds": cfg.MODEL.RPN.IOU_THRESHOLDS,
            "iou_labels": cfg.MODEL.RPN.IOU_LABELS,
            "head": build_rpn_head(cfg, shapes),
        }

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def label_and_sample_anchors(self, anchors_cat, gt_boxes_pad, num_gt):
        """rpn.py:307-363: IoU-match every anchor to the gt boxes of its image, then keep a random
        batch_size_per_image subset (the rest becomes -1).  Returns labels int8 [B, A], match int32 [B, A]."""
        assert self.anchor_boundary_thresh < 0
        match, labels, _ = F.iou_match(anchors_cat, gt_boxes_pad, num_gt, self.iou_thresholds[0], self.iou_thresholds[1], True)
        from . import sampling

        if sampling.permutation_source() is None:
            return self._subsample_batched(labels), match
        out = torch.full_like(labels, -1)
        for b in range(labels.shape[0]):
            lab = labels[b]
            pos_idx, neg_idx = subsample_labels(lab, self.batch_size_per_image, self.positive_fraction, 0)
            out[b, pos_idx] = 1
            out[b, neg_idx] = 0
        return out, match

    def _subsample_batched(self, labels):
        """sampling.py:38-54 for the whole batch without host synchronisation: one random key per anchor, the <= 128
        positives with the smallest keys are kept, the rest of the 256 is filled with the negatives with the smallest keys
        (= the reference's positive[randperm(P)[:num_pos]] with randperm := argsort of the keys).  Both draws share
        one u2_topk_rows_multi launch; the counts stay on the device."""
        from . import sampling

        b, a = labels.shape
        n = self.batch_size_per_image
        max_pos = int(n * self.positive_fraction)
        key = sampling.random_keys((b, a), labels.device)
        (_, pos_idx, pos_cnt), (_, neg_idx, neg_cnt) = F.topk_rows_multi([
            dict(vals=key, k=min(max_pos, a), largest=False, mask=labels, mask_value=1, want_vals=False),
            dict(vals=key, k=min(n, a), largest=False, mask=labels, mask_value=0, want_vals=False)])
        num_neg = torch.minimum(neg_cnt, n - pos_cnt)
        pos_valid = torch.arange(pos_idx.shape[1], device=labels.device)[None] < pos_cnt[:, None]
        neg_valid = torch.arange(neg_idx.shape[1], device=labels.device)[None] < num_neg[:, None]
        # entries beyond the counts are padding (index 0): they are redirected to a scratch column so that they can never
        # collide with a real pick of anchor 0 inside one scatter
        out = labels.new_full((b, a + 1), -1)
        out.scatter_(1, torch.where(neg_valid, neg_idx.long(), a), torch.zeros_like(neg_idx, dtype=out.dtype))
        out.scatter_(1, torch.where(pos_valid, pos_idx.long(), a), torch.ones_like(pos_idx, dtype=out.dtype))
        return out[:, :a].contiguous()

    def losses(self, anchors_per_level, objs, dlts, labels, match, gt_boxes_pad):
        b = labels.shape[0]
        normalizer = self.batch_size_per_image * b
        loss_cls, loss_loc = F.rpn_losses(labels, match, gt_boxes_pad, anchors_per_level, self.rpn_head.num_anchors,
                                          normalizer, objs, dlts)
        losses = {"loss_rpn_cls": loss_cls, "loss_rpn_loc": loss_loc}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

    @torch.no_grad()
    def predict_proposals(self, anchors_per_level, objs, dlts, image_sizes):
        """rpn.py:482-533 + proposal_utils.py:22-135: per-level top-k on the logits, decode only the selected
        anchors, clip, drop empty boxes, per-level NMS, keep the post_nms_topk best per image."""
        a = self.rpn_head.num_anchors
        b = objs[0].shape[0]
        dev = objs[0].device
        pre, post = self.pre_nms_topk[self.training], self.post_nms_topk[self.training]
        sizes = device_constant([list(x) for x in image_sizes], torch.float32, dev)
        # the k best logits of every image and level, ranked (logit descending, anchor index ascending), read straight from the
        # A valid columns of the 32-wide NHWC maps; all levels in one pair of launches
        tops = F.topk_rows_multi([dict(vals=o.contiguous(), k=min(o.shape[1] * o.shape[2] * a, pre), largest=True, group=a,
                                       pitch=o.shape[-1], n=o.shape[1] * o.shape[2] * a) for o in objs])
        kmax = max(min(o.shape[1] * o.shape[2] * a, pre) for o in objs)
        # Round 5: NMS per (image, level) row.  batched_nms never lets boxes of different levels meet (layers/nms.py:9-20 offsets
        # them apart), so the suppression decisions of a level only depend on that level's boxes in score order: the L lists of
        # <= PRE_NMS_TOPK boxes are L independent problems - L n^2 / 2 instead of (L n)^2 / 2 box pairs and scans of n / 64
        # instead of L n / 64 dependent steps (training: 0.80 -> 0.2 ms of kernels on the critical path) - and the survivors are
        # merged by score afterwards.  The merge ranks by (score descending, position in the level-major list ascending), the
        # same total order the single list was sorted in, so the proposals and their order are unchanged.
        # The rows themselves - deltas gathered from the NHWC maps, decoded against the anchors, clipped, short levels padded
        # (819 anchors at stride 64: zero boxes, score -3e38), finite / min-size filter - come from ONE launch over all levels
        # (F.rpn_decode; it was ~12 small launches per level and ~15 for the stacking and the masks).
        nl = len(objs)
        rows = b * nl
        boxes, scores, keep, nonfinite = F.rpn_decode(
            [dict(deltas=d, anchors=anc, idx=tops[lvl][1], scores=tops[lvl][0])
             for lvl, (anc, d) in enumerate(zip(anchors_per_level, dlts))],
            a, b, kmax, sizes, self.bbox_reg_weights, _SCALE_CLAMP, self.min_box_size)
        all_finite = nonfinite == 0  # raised as FloatingPointError when the counts are first read on the host
        # stable descending sort of every row's kept candidates by score (layers/nms.py:9-20 hands batched_nms score order)
        _, order, counts = F.topk_rows(scores, kmax, largest=True, mask=keep, mask_value=1, want_vals=False)
        order = order.long()
        s_boxes = torch.gather(boxes, 1, order[..., None].expand(-1, -1, 4)).contiguous()
        s_scores = torch.gather(scores, 1, order).contiguous()
        kept, nkeep = F.batched_nms(s_boxes, torch.zeros((rows, kmax), dtype=torch.int32, device=dev), counts, self.nms_thresh, kmax)
        # survivors as a mask over the sorted rows (positions >= nkeep of `kept` are padding that points at position 0: they add 0)
        alive = torch.zeros((rows, kmax), dtype=torch.int8, device=dev)
        alive.scatter_add_(1, kept.long(), (torch.arange(kmax, device=dev)[None] < nkeep[:, None]).to(torch.int8))
        # the post_nms_topk best survivors of an image over its L rows
        kpost = min(post, nl * kmax)
        _, idx, nkeep = F.topk_rows(s_scores.view(b, nl * kmax), kpost, largest=True, mask=alive.view(b, nl * kmax), mask_value=1,
                                    want_vals=False)
        idx = idx.long()  # rows >= nkeep[i] are zero: padding that points at a valid row
        p_boxes = torch.gather(s_boxes.view(b, nl * kmax, 4), 1, idx[..., None].expand(-1, -1, 4))
        p_scores = torch.gather(s_scores.view(b, nl * kmax), 1, idx)
        return LazyProposals(image_sizes, p_boxes, p_scores, nkeep, all_finite, self.training)

    def forward(self, image_sizes, features, gt_instances=None):
        feats = [features[f] for f in self.in_features]
        grid = [(f.shape[1], f.shape[2]) for f in feats]
        anchors_per_level = self.anchor_generator.grid_anchors(grid)
        objs, dlts = self.rpn_head(feats)
        if self.training:
            assert gt_instances is not None, "RPN requires gt_instances in training!"
            padded = PaddedTargets.of(gt_instances, objs[0].device)
            gt_pad, num_gt = padded.boxes, padded.counts
            anchors_cat = torch.cat(anchors_per_level, dim=0)
            labels, match = self.label_and_sample_anchors(anchors_cat, gt_pad, num_gt)
            losses = self.losses(anchors_per_level, objs, dlts, labels, match, gt_pad)
        else:
            losses = {}
        proposals = self.predict_proposals(anchors_per_level, [o.detach() for o in objs], [d.detach() for d in dlts],
                                           image_sizes)
        return proposals, losses


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)
