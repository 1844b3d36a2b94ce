"""CPU ORACLE - TEST INFRASTRUCTURE ONLY (see oracle/ops.py for the import rule).

Functional CPU restatement of the reference's PanopticFPN (cascade ROI heads) training forward and inference,
written as straight-line torch code over a name -> tensor parameter dictionary that uses the reference's
state-dict names.  fp32 by default (this mode is what is pinned against the reference in tests/golden);
``emulate_bf16=True`` rounds activations / weights to bfloat16 at the points where the reference's autocast (and the
HIP kernels) do, so the GPU path can be compared op-for-op at a tight tolerance.

Reference call graph followed (paths relative to the reference root):
  meta_arch/panoptic_fpn.py:90-181, meta_arch/rcnn.py:223-234, backbone/resnet.py:194-210,355-359,435-458,
  backbone/fpn.py:126-167,188-200, meta_arch/semantic_seg.py:231-267, proposal_generator/rpn.py:307-533,
  proposal_generator/proposal_utils.py:22-205, roi_heads/roi_heads.py:181-302,818-846,
  roi_heads/cascade_rcnn.py:137-299, roi_heads/box_head.py:94-97, roi_heads/fast_rcnn.py:118-171,288-463,
  roi_heads/mask_head.py:33-158, modeling/postprocessing.py:9-100.
"""
import math

import torch
import torch.nn.functional as F

from . import ops


class _ScaleGradient(torch.autograd.Function):
    """roi_heads/cascade_rcnn.py:20-28."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


class OracleModel:
    def __init__(self, cfg, state_dict, emulate_bf16=False, perm_fn=None, key_fn=None):
        self.cfg = cfg
        self.emulate = emulate_bf16
        self.perm_fn = perm_fn
        # key_fn(stage, image_index, n) -> n float keys: the sampling draws become argsort permutations of these keys
        # (ops.subsample_labels_keyed); stage is "rpn" or "roi"
        self.key_fn = key_fn
        self.p = {}
        for k, v in state_dict.items():
            t = v.detach().clone().float() if v.is_floating_point() else v.detach().clone()
            if t.is_floating_point() and not any(s in k for s in ("running_mean", "running_var")):
                t.requires_grad_(True)
            self.p[k] = t
        m = cfg.MODEL
        self.num_classes = m.ROI_HEADS.NUM_CLASSES
        self.sem_classes = m.SEM_SEG_HEAD.NUM_CLASSES
        self.pixel_mean = torch.tensor(m.PIXEL_MEAN).view(-1, 1, 1)
        self.pixel_std = torch.tensor(m.PIXEL_STD).view(-1, 1, 1)
        self.strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
        sizes, ars = m.ANCHOR_GENERATOR.SIZES, m.ANCHOR_GENERATOR.ASPECT_RATIOS
        ars = list(ars) * len(sizes) if len(ars) == 1 else ars
        self.cell_anchors = [ops.generate_cell_anchors(s, a).float() for s, a in zip(sizes, ars)]
        self.training = True

    @classmethod
    def from_config_file(cls, path, state_dict=None, opts=(), **kw):
        from u2seg_amd.config import get_cfg  # host-side config parsing only (no compute)

        cfg = get_cfg()
        cfg.merge_from_file(path)
        cfg.merge_from_list(["MODEL.DEVICE", "cpu"] + list(opts))
        if state_dict is None:
            from u2seg_amd.modeling import build_model  # module construction + init only (no forward)

            state_dict = build_model(cfg).state_dict()
        return cls(cfg, state_dict, **kw)

    def parameters(self):
        return {k: v for k, v in self.p.items() if v.requires_grad}

    # ---- primitives ---------------------------------------------------------------------------
    def q(self, x):
        return x.bfloat16().float() if self.emulate else x

    def conv(self, x, name, stride=1, pad=0, relu=False):
        w = self.q(self.p[name + ".weight"])
        b = self.p.get(name + ".bias")
        if b is not None:
            b = self.q(b)  # autocast casts every floating-point argument of the convolution, the bias included
        y = F.conv2d(self.q(x), w, b, stride, pad)
        if relu:
            y = F.relu(y)
        return self.q(y)

    def bn(self, x, name, residual=None, relu=False):
        if self.training:
            y = F.batch_norm(x, self.p[name + ".running_mean"], self.p[name + ".running_var"], self.p[name + ".weight"],
                             self.p[name + ".bias"], True, 0.1, 1e-5)
        else:
            y = F.batch_norm(x, self.p[name + ".running_mean"], self.p[name + ".running_var"], self.p[name + ".weight"],
                             self.p[name + ".bias"], False, 0.1, 1e-5)
        if residual is not None:
            # backbone/resnet.py:204-209 under autocast: the norm's output is a bf16 tensor, `out += shortcut` a bf16 add
            y = self.q(y) + residual
        if relu:
            y = F.relu(y)
        return self.q(y)

    def gn(self, x, name, relu=True):
        y = F.group_norm(x, 32, self.p[name + ".weight"], self.p[name + ".bias"], 1e-5)
        return self.q(F.relu(y) if relu else y)

    def linear(self, x, name, relu=False):
        y = F.linear(self.q(x), self.q(self.p[name + ".weight"]), self.q(self.p[name + ".bias"]))
        if relu:
            y = F.relu(y)
        return self.q(y)

    # ---- backbone -----------------------------------------------------------------------------
    def stem(self, images):
        """backbone/resnet.py:355-359 (BasicStem: conv 7x7/2 + norm + relu_, max_pool2d 3x3/2)."""
        pre = "backbone.bottom_up."
        x = self.bn(self.conv(images, pre + "stem.conv1", 2, 3), pre + "stem.conv1.norm", relu=True)
        return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)

    def bottleneck(self, x, si, bi, capture=None):
        """backbone/resnet.py:194-210 (BottleneckBlock.forward, stride on the 3x3 conv: STRIDE_IN_1X1 False)."""
        n = "backbone.bottom_up.res%d.%d." % (si, bi)
        stride = 2 if (bi == 0 and si > 2) else 1
        c1 = self.bn(self.conv(x, n + "conv1"), n + "conv1.norm", relu=True)
        c2 = self.bn(self.conv(c1, n + "conv2", stride, 1), n + "conv2.norm", relu=True)
        out = self.conv(c2, n + "conv3")
        if (n + "shortcut.weight") in self.p:
            sc = self.bn(self.conv(x, n + "shortcut", stride), n + "shortcut.norm")
        else:
            sc = x
        y = self.bn(out, n + "conv3.norm", residual=sc, relu=True)
        if capture is not None:  # the units inside the block, for per-layer teacher forcing
            capture["res%d.%d.conv1" % (si, bi)], capture["res%d.%d.conv2" % (si, bi)] = c1, c2
            capture["res%d.%d.shortcut" % (si, bi)] = sc
            capture["res%d.%d" % (si, bi)] = y
        return y

    def backbone(self, images, capture=None):
        x = self.stem(images)
        if capture is not None:
            capture["stem"] = x
        res = {}
        for si, nblocks in zip(range(2, 6), [3, 4, 6, 3]):
            for bi in range(nblocks):
                x = self.bottleneck(x, si, bi, capture)
            res["res%d" % si] = x
        return self.fpn(res)

    def fpn(self, res):
        """fpn.py:126-167."""
        feats = {}
        prev = self.bn(self.conv(res["res5"], "backbone.fpn_lateral5"), "backbone.fpn_lateral5.norm")
        feats["p5"] = self.bn(self.conv(prev, "backbone.fpn_output5", 1, 1), "backbone.fpn_output5.norm")
        for lvl in (4, 3, 2):
            lat = self.bn(self.conv(res["res%d" % lvl], "backbone.fpn_lateral%d" % lvl), "backbone.fpn_lateral%d.norm" % lvl)
            prev = self.q(lat + F.interpolate(prev, scale_factor=2.0, mode="nearest"))
            feats["p%d" % lvl] = self.bn(self.conv(prev, "backbone.fpn_output%d" % lvl, 1, 1), "backbone.fpn_output%d.norm" % lvl)
        feats["p6"] = F.max_pool2d(feats["p5"], kernel_size=1, stride=2, padding=0)
        return feats

    def preprocess(self, batched_inputs, div=32):
        """rcnn.py:223-234 + structures/image_list.py:59-129."""
        imgs = [(x["image"].float() - self.pixel_mean) / self.pixel_std for x in batched_inputs]
        sizes = [(im.shape[-2], im.shape[-1]) for im in imgs]
        mh = (max(s[0] for s in sizes) + div - 1) // div * div
        mw = (max(s[1] for s in sizes) + div - 1) // div * div
        out = torch.zeros((len(imgs), 3, mh, mw))
        for i, im in enumerate(imgs):
            out[i, :, : im.shape[1], : im.shape[2]] = im
        return out, sizes, (mh, mw)

    # ---- semantic head ------------------------------------------------------------------------
    def sem_seg_logits(self, feats, capture=None):
        x = None
        for f, stride in (("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32)):
            y = feats[f]
            head_len = max(1, int(math.log2(stride) - math.log2(4)))
            idx = 0
            for _ in range(head_len):
                n = "sem_seg_head.%s.%d" % (f, idx)
                y = self.gn(self.conv(y, n, 1, 1), n + ".norm")
                if capture is not None:
                    capture[n] = y
                idx += 1
                if stride != 4:
                    y = self.q(F.interpolate(y, scale_factor=2.0, mode="bilinear", align_corners=False))
                    idx += 1
            x = y if x is None else self.q(x + y)
        if capture is not None:
            capture["sem_seg_head.sum"] = x
        return self.conv(x, "sem_seg_head.predictor")

    def sem_seg_loss(self, logits, targets):
        pred = F.interpolate(logits.float(), scale_factor=4.0, mode="bilinear", align_corners=False)
        return F.cross_entropy(pred, targets, reduction="mean", ignore_index=self.cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE) * \
            self.cfg.MODEL.SEM_SEG_HEAD.LOSS_WEIGHT

    # ---- RPN ----------------------------------------------------------------------------------
    def rpn_head(self, feats):
        objs, dlts = [], []
        for f in self.cfg.MODEL.RPN.IN_FEATURES:
            t = self.conv(feats[f], "proposal_generator.rpn_head.conv", 1, 1, relu=True)
            o = self.conv(t, "proposal_generator.rpn_head.objectness_logits")
            d = self.conv(t, "proposal_generator.rpn_head.anchor_deltas")
            n = o.shape[0]
            objs.append(o.permute(0, 2, 3, 1).flatten(1))
            dlts.append(d.view(n, -1, 4, d.shape[-2], d.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2))
        return objs, dlts

    def anchors(self, feats):
        fs = self.cfg.MODEL.RPN.IN_FEATURES
        grid = [(feats[f].shape[-2], feats[f].shape[-1]) for f in fs]
        return ops.grid_anchors(grid, [self.strides[f] for f in fs], self.cell_anchors, self.cfg.MODEL.ANCHOR_GENERATOR.OFFSET)

    @torch.no_grad()
    def rpn_label_and_sample(self, anchors_cat, gt_instances):
        r = self.cfg.MODEL.RPN
        labels, matched = [], []
        for i, inst in enumerate(gt_instances):
            gtb = inst.gt_boxes.tensor
            mq = ops.pairwise_iou(gtb, anchors_cat)
            midx, lab = ops.matcher(mq, r.IOU_THRESHOLDS, r.IOU_LABELS, True)
            if self.key_fn is not None:
                pos, neg = ops.subsample_labels_keyed(lab, r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION, 0,
                                                      self.key_fn("rpn", i, lab.numel()))
            else:
                pos, neg = ops.subsample_labels(lab, r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION, 0, self.perm_fn)
            lab.fill_(-1)
            lab.scatter_(0, pos, 1)
            lab.scatter_(0, neg, 0)
            labels.append(lab)
            matched.append(gtb[midx] if len(gtb) else torch.zeros_like(anchors_cat))
        return labels, matched

    def rpn_losses(self, anchors, objs, dlts, labels, matched):
        r = self.cfg.MODEL.RPN
        n = len(labels)
        gl = torch.stack(labels)
        pos = gl == 1
        anchors_cat = torch.cat(anchors)
        gt_deltas = torch.stack([ops.get_deltas(anchors_cat, m, r.BBOX_REG_WEIGHTS) for m in matched])
        pd = torch.cat(dlts, dim=1).float()
        loc = torch.abs(pd[pos] - gt_deltas[pos]).sum()  # smooth_l1 with beta 0 == L1 (box_regression.py:340-345)
        valid = gl >= 0
        cls = F.binary_cross_entropy_with_logits(torch.cat(objs, dim=1).float()[valid], gl[valid].float(), reduction="sum")
        norm = r.BATCH_SIZE_PER_IMAGE * n
        return {"loss_rpn_cls": cls / norm * r.LOSS_WEIGHT, "loss_rpn_loc": loc / norm * r.BBOX_REG_LOSS_WEIGHT * r.LOSS_WEIGHT}

    @torch.no_grad()
    def rpn_proposals(self, anchors, objs, dlts, image_sizes):
        """rpn.py:482-533 + proposal_utils.py:22-135."""
        r = self.cfg.MODEL.RPN
        pre = r.PRE_NMS_TOPK_TRAIN if self.training else r.PRE_NMS_TOPK_TEST
        post = r.POST_NMS_TOPK_TRAIN if self.training else r.POST_NMS_TOPK_TEST
        n = objs[0].shape[0]
        props = []
        for a, d in zip(anchors, dlts):
            bsz = d.shape[0]
            dd = d.reshape(-1, 4)
            aa = a.unsqueeze(0).expand(bsz, -1, -1).reshape(-1, 4)
            props.append(ops.apply_deltas(dd, aa, r.BBOX_REG_WEIGHTS).view(bsz, -1, 4))
        tops, topp, lvl = [], [], []
        bi = torch.arange(n)
        for li, (p, lg) in enumerate(zip(props, objs)):
            k = min(lg.shape[1], pre)
            # proposal_utils.py:79-80 calls topk, whose order among equal logits is unspecified; the oracle fixes it the way a
            # stable sort does (logit descending, anchor index ascending) - the rule the HIP selection kernel implements
            srt = torch.sort(lg, dim=1, descending=True, stable=True)
            ts, ti = srt[0][:, :k], srt[1][:, :k]
            tops.append(ts)
            topp.append(p[bi[:, None], ti])
            lvl.append(torch.full((k,), li, dtype=torch.int64))
        tops, topp, lvl = torch.cat(tops, 1), torch.cat(topp, 1), torch.cat(lvl, 0)
        out = []
        for i, size in enumerate(image_sizes):
            boxes, sc, lv = topp[i], tops[i].float(), lvl
            valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(sc)
            if not valid.all():
                if self.training:
                    raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
                boxes, sc, lv = boxes[valid], sc[valid], lv[valid]
            boxes = ops.clip_boxes(boxes, size)
            keep = ops.nonempty(boxes, self.cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
            boxes, sc, lv = boxes[keep], sc[keep], lv[keep]
            keep = ops.nms(boxes, sc, r.NMS_THRESH, lv)[:post]
            out.append({"proposal_boxes": boxes[keep], "objectness_logits": sc[keep], "image_size": size})
        return out

    # ---- ROI heads ----------------------------------------------------------------------------
    def _match(self, boxes, gt_boxes, thr):
        mq = ops.pairwise_iou(gt_boxes, boxes)
        return ops.matcher(mq, [thr], [0, 1], False)

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, gt_instances):
        """roi_heads.py:220-302 (+ add_ground_truth_to_proposals, proposal_utils.py:138-205)."""
        h = self.cfg.MODEL.ROI_HEADS
        K = self.num_classes
        out = []
        gt_logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
        for i, (prop, inst) in enumerate(zip(proposals, gt_instances)):
            gtb, gtc = inst.gt_boxes.tensor, inst.gt_classes
            boxes = torch.cat([prop["proposal_boxes"], gtb])
            logits = torch.cat([prop["objectness_logits"], gt_logit * torch.ones(len(gtb))])
            midx, mlab = self._match(boxes, gtb, h.IOU_THRESHOLDS[0])
            if len(gtb) > 0:
                cls = gtc[midx]
                cls[mlab == 0] = K
                cls[mlab == -1] = -1
            else:
                cls = torch.zeros_like(midx) + K
            if self.key_fn is not None:
                fg, bg = ops.subsample_labels_keyed(cls, h.BATCH_SIZE_PER_IMAGE, h.POSITIVE_FRACTION, K,
                                                    self.key_fn("roi", i, cls.numel()))
            else:
                fg, bg = ops.subsample_labels(cls, h.BATCH_SIZE_PER_IMAGE, h.POSITIVE_FRACTION, K, self.perm_fn)
            sel = torch.cat([fg, bg])
            res = {"proposal_boxes": boxes[sel], "objectness_logits": logits[sel], "gt_classes": cls[sel],
                   "image_size": prop["image_size"]}
            if len(gtb) > 0:
                st = midx[sel]
                res["gt_boxes"] = gtb[st]
                if inst.has("gt_masks"):
                    res["gt_masks"] = inst.gt_masks.tensor[st]
            out.append(res)
        return out

    def box_head(self, x, k):
        x = x.flatten(1)
        x = self.linear(x, "roi_heads.box_head.%d.fc1" % k, relu=True)
        return self.linear(x, "roi_heads.box_head.%d.fc2" % k, relu=True)

    def run_stage(self, feat_list, proposals, k):
        scales = [1.0 / self.strides[f] for f in self.cfg.MODEL.ROI_HEADS.IN_FEATURES]
        pooled = ops.roi_pool_multilevel(feat_list, [p["proposal_boxes"] for p in proposals],
                                         self.cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION, scales)
        pooled = self.q(pooled)
        if self.training:
            pooled = _ScaleGradient.apply(pooled, 1.0 / 3)
        x = self.box_head(pooled, k)
        scores = self.linear(x, "roi_heads.box_predictor.%d.cls_score" % k)
        deltas = self.linear(x, "roi_heads.box_predictor.%d.bbox_pred" % k)
        return scores, deltas

    def box_losses(self, scores, deltas, proposals, weights):
        gt_classes = torch.cat([p["gt_classes"] for p in proposals])
        K = self.num_classes
        loss_cls = F.cross_entropy(scores.float(), gt_classes, reduction="mean")
        boxes = torch.cat([p["proposal_boxes"] for p in proposals])
        gtb = torch.cat([p.get("gt_boxes", p["proposal_boxes"]) for p in proposals])
        fg = torch.nonzero((gt_classes >= 0) & (gt_classes < K), as_tuple=True)[0]
        tgt = ops.get_deltas(boxes[fg], gtb[fg], weights)
        pred = deltas.float()
        if pred.shape[1] != 4:  # class-specific regression (fast_rcnn.py:434-437): the 4 outputs of the gt class
            pred = pred.view(-1, K, 4)[fg, gt_classes[fg]]
        else:
            pred = pred[fg]
        loss_box = torch.abs(pred - tgt).sum() / max(gt_classes.numel(), 1.0)
        return loss_cls, loss_box

    @torch.no_grad()
    def cascade_next_stage(self, prev_boxes, image_sizes, gt_instances, k):
        """cascade_rcnn.py:226-299: the boxes stage k-1 predicted become stage k's proposals - detached, clipped, empty ones
        dropped in training (_create_proposals_from_boxes), then matched at IOUS[k] and labelled (_match_and_label_boxes)."""
        ious = self.cfg.MODEL.ROI_BOX_CASCADE_HEAD.IOUS
        K = self.num_classes
        out = []
        for b, size in zip(prev_boxes, image_sizes):
            bx = ops.clip_boxes(b.detach(), size)
            if self.training:
                bx = bx[ops.nonempty(bx)]
            out.append({"proposal_boxes": bx, "image_size": size})
        if self.training:
            for p, inst in zip(out, gt_instances):
                gtb = inst.gt_boxes.tensor
                midx, lab = self._match(p["proposal_boxes"], gtb, ious[k])
                if len(gtb) > 0:
                    cls = inst.gt_classes[midx]
                    cls[lab == 0] = K
                    p["gt_classes"], p["gt_boxes"] = cls, gtb[midx]
                else:
                    p["gt_classes"] = torch.zeros_like(midx) + K
                    p["gt_boxes"] = torch.zeros((len(p["proposal_boxes"]), 4))
        return out

    def forward_box(self, feats, proposals, gt_instances):
        m = self.cfg.MODEL
        feat_list = [feats[f] for f in m.ROI_HEADS.IN_FEATURES]
        weights, ious = m.ROI_BOX_CASCADE_HEAD.BBOX_REG_WEIGHTS, m.ROI_BOX_CASCADE_HEAD.IOUS
        K = self.num_classes
        outs = []
        prev = None
        for k in range(3):
            if k > 0:
                proposals = self.cascade_next_stage(prev, [p["image_size"] for p in proposals], gt_instances, k)
            scores, deltas = self.run_stage(feat_list, proposals, k)
            boxes = torch.cat([p["proposal_boxes"] for p in proposals])
            pred = ops.apply_deltas(deltas, boxes, weights[k])
            prev = pred.split([len(p["proposal_boxes"]) for p in proposals])
            outs.append((scores, deltas, proposals))
        return outs

    def mask_head_features(self, x):
        i = 1
        while "roi_heads.mask_head.mask_fcn%d.weight" % i in self.p:  # ROI_MASK_HEAD.NUM_CONV layers (4 in the U2Seg configs)
            x = self.conv(x, "roi_heads.mask_head.mask_fcn%d" % i, 1, 1, relu=True)
            i += 1
        w, b = self.q(self.p["roi_heads.mask_head.deconv.weight"]), self.p["roi_heads.mask_head.deconv.bias"]
        return self.q(F.relu(F.conv_transpose2d(self.q(x), w, b, stride=2)))

    def mask_loss(self, feats, proposals):
        K = self.num_classes
        scales = [1.0 / self.strides[f] for f in self.cfg.MODEL.ROI_HEADS.IN_FEATURES]
        fgs = []
        for p in proposals:
            sel = torch.nonzero((p["gt_classes"] != -1) & (p["gt_classes"] != K), as_tuple=True)[0]
            fgs.append({k: (v[sel] if torch.is_tensor(v) else v) for k, v in p.items()})
        feat_list = [feats[f] for f in self.cfg.MODEL.ROI_HEADS.IN_FEATURES]
        pooled = self.q(ops.roi_pool_multilevel(feat_list, [p["proposal_boxes"] for p in fgs],
                                                self.cfg.MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION, scales))
        x = self.mask_head_features(pooled)
        logits = self.conv(x, "roi_heads.mask_head.predictor")
        return self.mask_loss_from_logits(logits, fgs)

    def mask_loss_from_logits(self, logits, fgs):
        """mask_rcnn_loss (mask_head.py:33-112): BCE between the gt-class channel of every foreground ROI and its gt mask
        cropped to the ROI at the logits' resolution."""
        side = logits.shape[-1]
        gt_classes, gt_masks = [], []
        for p in fgs:
            if len(p["proposal_boxes"]) == 0:
                continue
            gt_classes.append(p["gt_classes"].to(torch.int64))
            gt_masks.append(ops.crop_and_resize_masks(p["gt_masks"], p["proposal_boxes"], side))
        if len(gt_masks) == 0:
            return logits.sum() * 0
        gt_classes, gt_masks = torch.cat(gt_classes), torch.cat(gt_masks)
        sel = logits[torch.arange(logits.shape[0]), gt_classes]
        return F.binary_cross_entropy_with_logits(sel.float(), gt_masks.float(), reduction="mean")

    # ---- top level ----------------------------------------------------------------------------
    def train_forward(self, batched_inputs, return_internals=False):
        """panoptic_fpn.py:90-138 -> dict of the 10 losses."""
        self.training = True
        images, sizes, (mh, mw) = self.preprocess(batched_inputs)
        feats = self.backbone(images)
        ign = self.cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE
        tgt = torch.full((len(batched_inputs), mh, mw), ign, dtype=torch.int64)
        for i, x in enumerate(batched_inputs):
            s = x["sem_seg"]
            tgt[i, : s.shape[0], : s.shape[1]] = s
        losses = {"loss_sem_seg": self.sem_seg_loss(self.sem_seg_logits(feats), tgt)}
        gt_instances = [x["instances"] for x in batched_inputs]
        anchors = self.anchors(feats)
        objs, dlts = self.rpn_head(feats)
        labels, matched = self.rpn_label_and_sample(torch.cat(anchors), gt_instances)
        losses.update(self.rpn_losses(anchors, objs, dlts, labels, matched))
        proposals = self.rpn_proposals(anchors, [o.detach() for o in objs], [d.detach() for d in dlts], sizes)
        sampled = self.label_and_sample_proposals(proposals, gt_instances)
        outs = self.forward_box(feats, sampled, gt_instances)
        w = self.cfg.MODEL.ROI_BOX_CASCADE_HEAD.BBOX_REG_WEIGHTS
        for k, (scores, deltas, props) in enumerate(outs):
            lc, lb = self.box_losses(scores, deltas, props, w[k])
            losses["loss_cls_stage%d" % k] = lc
            losses["loss_box_reg_stage%d" % k] = lb * self.cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT
        losses["loss_mask"] = self.mask_loss(feats, sampled)
        if return_internals:
            return losses, {"feats": feats, "proposals": proposals, "sampled": sampled, "rpn_labels": labels,
                            "objs": objs, "dlts": dlts}
        return losses

    # ---- inference (panoptic_fpn.py:140-181) ---------------------------------------------------------
    @staticmethod
    def paste_masks(masks, boxes, image_shape, threshold=0.5):
        """layers/mask_ops.py:17-147, full-image grid_sample form (the reference's GPU branch; its CPU branch
        only skips empty regions and yields the same values)."""
        n = masks.shape[0]
        img_h, img_w = image_shape
        if n == 0:
            return torch.zeros((0, img_h, img_w), dtype=torch.bool)
        x0, y0, x1, y1 = torch.split(boxes, 1, dim=1)
        img_y = torch.arange(0, img_h, dtype=torch.float32) + 0.5
        img_x = torch.arange(0, img_w, dtype=torch.float32) + 0.5
        img_y = (img_y - y0) / (y1 - y0) * 2 - 1
        img_x = (img_x - x0) / (x1 - x0) * 2 - 1
        gx = img_x[:, None, :].expand(n, img_y.size(1), img_x.size(1))
        gy = img_y[:, :, None].expand(n, img_y.size(1), img_x.size(1))
        grid = torch.stack([gx, gy], dim=3)
        img = F.grid_sample(masks[:, None].float(), grid, align_corners=False)
        return img[:, 0] >= threshold

    @staticmethod
    def combine_panoptic(masks, scores, classes, sem, overlap_thr, stuff_area, inst_thr):
        """meta_arch/panoptic_fpn.py:184-269."""
        pan = torch.zeros_like(sem, dtype=torch.int32)
        order = torch.argsort(-scores)
        seg_id, info = 0, []
        for i in order.tolist():
            score = float(scores[i])
            if score < inst_thr:
                break
            m = masks[i]
            area = int(m.sum())
            if area == 0:
                continue
            inter = int((m & (pan > 0)).sum())
            if inter * 1.0 / area > overlap_thr:
                continue
            if inter > 0:
                m = m & (pan == 0)
            seg_id += 1
            pan[m] = seg_id
            info.append({"id": seg_id, "isthing": True, "score": score, "category_id": int(classes[i]), "instance_id": i})
        for label in torch.unique(sem).tolist():
            if label == 0:
                continue
            m = (sem == label) & (pan == 0)
            area = int(m.sum())
            if area < stuff_area:
                continue
            seg_id += 1
            pan[m] = seg_id
            info.append({"id": seg_id, "isthing": False, "category_id": label, "area": area})
        return pan, info

    def box_inference_single(self, boxes, scores, image_size):
        """roi_heads/fast_rcnn.py:118-171 (class-agnostic boxes)."""
        h = self.cfg.MODEL.ROI_HEADS
        valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
        boxes, scores = boxes[valid], scores[valid]
        scores = scores[:, :-1]
        boxes = ops.clip_boxes(boxes, image_size)
        mask = scores > h.SCORE_THRESH_TEST
        inds = mask.nonzero()
        b, s = boxes[inds[:, 0]], scores[mask]
        keep = ops.nms(b, s, h.NMS_THRESH_TEST, inds[:, 1])[: self.cfg.TEST.DETECTIONS_PER_IMAGE]
        return b[keep], s[keep], inds[keep, 1]

    @torch.no_grad()
    def inference(self, batched_inputs):
        self.training = False
        m = self.cfg.MODEL
        images, sizes, _ = self.preprocess(batched_inputs)
        feats = self.backbone(images)
        sem = F.interpolate(self.sem_seg_logits(feats).float(), scale_factor=4.0, mode="bilinear", align_corners=False)
        anchors = self.anchors(feats)
        objs, dlts = self.rpn_head(feats)
        proposals = self.rpn_proposals(anchors, objs, dlts, sizes)
        outs = self.forward_box(feats, proposals, None)
        probs = sum(F.softmax(o[0].float(), dim=-1) for o in outs) * (1.0 / 3)
        scores, deltas, props = outs[-1]
        boxes = ops.apply_deltas(deltas, torch.cat([p["proposal_boxes"] for p in props]), m.ROI_BOX_CASCADE_HEAD.BBOX_REG_WEIGHTS[2])
        counts = [len(p["proposal_boxes"]) for p in props]
        results = []
        scales = [1.0 / self.strides[f] for f in m.ROI_HEADS.IN_FEATURES]
        feat_list = [feats[f] for f in m.ROI_HEADS.IN_FEATURES]
        dets = [self.box_inference_single(b, s, sz) for b, s, sz in zip(boxes.split(counts), probs.split(counts), sizes)]
        pooled = self.q(ops.roi_pool_multilevel(feat_list, [d[0] for d in dets], m.ROI_MASK_HEAD.POOLER_RESOLUTION, scales))
        logits = self.conv(self.mask_head_features(pooled), "roi_heads.mask_head.predictor") if pooled.shape[0] else pooled
        off = 0
        for i, ((b, s, c), size, inp) in enumerate(zip(dets, sizes, batched_inputs)):
            n = b.shape[0]
            mp = logits[off : off + n][torch.arange(n), c].float().sigmoid() if n else torch.zeros((0, 28, 28))
            off += n
            oh, ow = inp.get("height", size[0]), inp.get("width", size[1])
            sem_r = F.interpolate(sem[i, :, : size[0], : size[1]][None], size=(oh, ow), mode="bilinear", align_corners=False)[0]
            bs = b.clone()
            bs[:, 0::2] *= ow / size[1]
            bs[:, 1::2] *= oh / size[0]
            bs = ops.clip_boxes(bs, (oh, ow))
            ne = ops.nonempty(bs)
            bs, s2, c2, mp = bs[ne], s[ne], c[ne], mp[ne]
            masks = self.paste_masks(mp, bs, (oh, ow))
            pc = m.PANOPTIC_FPN.COMBINE
            pan = self.combine_panoptic(masks, s2, c2, sem_r.argmax(dim=0), pc.OVERLAP_THRESH, pc.STUFF_AREA_LIMIT,
                                        pc.INSTANCES_CONFIDENCE_THRESH)
            results.append({"sem_seg": sem_r, "boxes": bs, "scores": s2, "classes": c2, "masks": masks, "panoptic_seg": pan})
        return results
