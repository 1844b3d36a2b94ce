"""Meta-architectures: GeneralizedRCNN and PanopticFPN with detectron2's forward contract
(detectron2/modeling/meta_arch/rcnn.py:25-234, meta_arch/panoptic_fpn.py:21-181, meta_arch/build.py:7-25)."""
import os

import torch
from torch import nn

from ..config import configurable
from ..layers import functional as F
from ..structures import ImageList
from ..utils.registry import Registry
from .backbone import build_backbone
from .inference import (combine_semantic_and_instance_outputs, combine_semantic_and_instance_outputs_batch,
                        detector_postprocess, detector_postprocess_batch, sem_seg_postprocess)
from .roi_heads import build_roi_heads
from .rpn import build_proposal_generator
from .semantic_seg import build_sem_seg_head

META_ARCH_REGISTRY = Registry("META_ARCH")


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN(nn.Module):
    @configurable
    def __init__(self, *, backbone, proposal_generator, roi_heads, pixel_mean, pixel_std, input_format=None, vis_period=0):
        super().__init__()
        self.backbone, self.proposal_generator, self.roi_heads = backbone, proposal_generator, roi_heads
        self.input_format, self.vis_period = input_format, vis_period
        self.on_heads_backward_done = None  # optional callback (engine/trainer.py): all head gradients are final
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std).view(-1, 1, 1), False)
        assert self.pixel_mean.shape == self.pixel_std.shape

    @classmethod
    def from_config(cls, cfg):
        backbone = build_backbone(cfg)
        return {
            "backbone": backbone,
            "proposal_generator": build_proposal_generator(cfg, backbone.output_shape()),
            "roi_heads": build_roi_heads(cfg, backbone.output_shape()),
            "input_format": cfg.INPUT.FORMAT,
            "vis_period": cfg.VIS_PERIOD,
            "pixel_mean": cfg.MODEL.PIXEL_MEAN,
            "pixel_std": cfg.MODEL.PIXEL_STD,
        }

    @property
    def device(self):
        return self.pixel_mean.device

    @staticmethod
    def _fan_out_features(features, consumers):
        """One feature dict per consumer (`consumers`: their in_features lists).  A map read by several of them gets one autograd
        handle per reader, so that its gradient is summed by one kernel (layers/functional.py:fan_out)."""
        readers = {name: [i for i, names in enumerate(consumers) if name in names] for name in features}
        out = [dict(features) for _ in consumers]
        for name, who in readers.items():
            if len(who) > 1:
                for i, handle in zip(who, F.fan_out(features[name], len(who))):
                    out[i][name] = handle
        return out

    def _watch_feature_grads(self, features):
        """Fire on_heads_backward_done once the gradient of every FPN output has been computed in backward."""
        cb = self.on_heads_backward_done
        if cb is None or not torch.is_grad_enabled():
            return
        watched = [f for f in features.values() if f.requires_grad]
        state = {"left": len(watched)}

        def hook(_grad):
            state["left"] -= 1
            if state["left"] == 0:
                cb()

        for f in watched:
            f.register_hook(hook)

    def _backbone_features(self, batched_inputs):
        """rcnn.py:223-234 + backbone: the stem kernel normalises, pads and convolves in one pass."""
        images = [x["image"].to(self.device).contiguous() for x in batched_inputs]
        image_sizes = [(im.shape[-2], im.shape[-1]) for im in images]
        padded_hw = ImageList.padded_size(image_sizes, self.backbone.size_divisibility)
        mean = self.pixel_mean.view(-1).float().contiguous()
        std = self.pixel_std.view(-1).float().contiguous()
        return self.backbone(images, mean, std, padded_hw), image_sizes, padded_hw

    def forward(self, batched_inputs):
        if not self.training:
            return self.inference(batched_inputs)
        features, image_sizes, _ = self._backbone_features(batched_inputs)
        self._watch_feature_grads(features)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        rpn_f, roi_f = self._fan_out_features(features, [self.proposal_generator.in_features, self.roi_heads.box_in_features])
        proposals, proposal_losses = self.proposal_generator(image_sizes, rpn_f, gt_instances)
        _, detector_losses = self.roi_heads(None, roi_f, proposals, gt_instances)
        F.join_aux_stream(self.device)  # the RPN losses were computed on the second stream
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses

    def inference(self, batched_inputs, do_postprocess=True):
        assert not self.training
        features, image_sizes, _ = self._backbone_features(batched_inputs)
        proposals, _ = self.proposal_generator(image_sizes, features, None)
        results, _ = self.roi_heads(None, features, proposals, None)
        if not do_postprocess:
            return results
        out_sizes = [(inp.get("height", size[0]), inp.get("width", size[1])) for inp, size in zip(batched_inputs, image_sizes)]
        return [{"instances": r} for r in detector_postprocess_batch(results, out_sizes)]


@META_ARCH_REGISTRY.register()
class PanopticFPN(GeneralizedRCNN):
    @configurable
    def __init__(self, *, sem_seg_head, combine_overlap_thresh=0.5, combine_stuff_area_thresh=4096,
                 combine_instances_score_thresh=0.5, **kwargs):
        super().__init__(**kwargs)
        self.sem_seg_head = sem_seg_head
        self.combine_overlap_thresh = combine_overlap_thresh
        self.combine_stuff_area_thresh = combine_stuff_area_thresh
        self.combine_instances_score_thresh = combine_instances_score_thresh

    @classmethod
    def from_config(cls, cfg):
        ret = super().from_config(cfg)
        ret.update({
            "combine_overlap_thresh": cfg.MODEL.PANOPTIC_FPN.COMBINE.OVERLAP_THRESH,
            "combine_stuff_area_thresh": cfg.MODEL.PANOPTIC_FPN.COMBINE.STUFF_AREA_LIMIT,
            "combine_instances_score_thresh": cfg.MODEL.PANOPTIC_FPN.COMBINE.INSTANCES_CONFIDENCE_THRESH,
        })
        ret["sem_seg_head"] = build_sem_seg_head(cfg, ret["backbone"].output_shape())
        if cfg.MODEL.PANOPTIC_FPN.INSTANCE_LOSS_WEIGHT != 1.0:
            w = cfg.MODEL.PANOPTIC_FPN.INSTANCE_LOSS_WEIGHT

            def update_weight(x):
                return {k: v * w for k, v in x.items()} if isinstance(x, dict) else x * w

            roi_heads = ret["roi_heads"]
            for p in roi_heads.box_predictor if isinstance(roi_heads.box_predictor, nn.ModuleList) else [roi_heads.box_predictor]:
                p.loss_weight = update_weight(p.loss_weight)
            roi_heads.mask_head.loss_weight = update_weight(roi_heads.mask_head.loss_weight)
        return ret

    def _sem_seg_targets(self, batched_inputs, padded_hw):
        """ImageList.from_tensors(gt_sem_seg, size_divisibility, ignore_value) (panoptic_fpn.py:118-126) as uint8."""
        ignore = self.sem_seg_head.ignore_value
        b = len(batched_inputs)
        maps = [x["sem_seg"].to(self.device) for x in batched_inputs]
        if maps[0].is_cuda and padded_hw[1] % 16 == 0 and all(m.dtype == maps[0].dtype for m in maps) \
                and maps[0].dtype in (torch.int64, torch.uint8):
            # one launch for the batch (u2_label_pad_batch): per image it was a conversion and a strided copy
            import ctypes

            from .. import _hip

            maps = [m.contiguous() for m in maps]
            out = torch.empty((b, padded_hw[0], padded_hw[1]), dtype=torch.uint8, device=self.device)
            ptrs = (ctypes.c_void_p * b)(*[m.data_ptr() for m in maps])
            hs = (ctypes.c_int * b)(*[m.shape[0] for m in maps])
            ws = (ctypes.c_int * b)(*[m.shape[1] for m in maps])
            _hip.call("u2_label_pad_batch", ptrs, hs, ws, b, int(maps[0].dtype == torch.int64), out, padded_hw[0], padded_hw[1],
                      int(ignore))
            return out
        out = torch.full((b, padded_hw[0], padded_hw[1]), ignore, dtype=torch.uint8, device=self.device)
        for i, t in enumerate(maps):
            out[i, : t.shape[0], : t.shape[1]] = t.to(torch.uint8)
        return out

    def forward(self, batched_inputs):
        """panoptic_fpn.py:90-138: training returns the 10 loss entries of the cascade Panoptic-FPN."""
        if not self.training:
            return self.inference(batched_inputs)
        features, image_sizes, padded_hw = self._backbone_features(batched_inputs)
        self._watch_feature_grads(features)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        assert "sem_seg" in batched_inputs[0]
        gt_sem_seg = self._sem_seg_targets(batched_inputs, padded_hw)
        sem_f, rpn_f, roi_f = self._fan_out_features(
            features, [self.sem_seg_head.in_features, self.proposal_generator.in_features, self.roi_heads.box_in_features])
        # The semantic head only shares the FPN maps with the detector branch: it runs on a second stream, so that its large
        # kernels fill the chip while the proposal / sampling bookkeeping of the other branch occupies a few CUs at a time
        # (autograd replays every node on the stream of its forward pass, so the backward passes overlap the same way).
        aux = F.aux_stream(self.device) if sem_f[self.sem_seg_head.in_features[0]].is_cuda else None
        F.clear_deferred()
        if aux is not None and os.environ.get("U2_SEM_PIECES", "1") != "0":
            # Round 6: the head is handed over in pieces (level stacks, then predictor + loss) that the ROI heads launch - on `aux` -
            # where their own chain is about to be host-bound: in front of the sampler's host synchronisation, between the cascade
            # stages, in front of the losses (layers/functional.py:defer_pieces).  Launched as a whole in front of the RPN (rounds
            # 2-5) it ran beside the RPN's convolutions and was finished when the idle stretches began (tools/step_idle.py).
            sem_seg_losses = {}
            main = torch.cuda.current_stream(self.device)
            first = [True]

            def on_aux(piece):
                def run():
                    if first[0]:
                        aux.wait_stream(main)   # the FPN maps and the targets exist (nothing later on `main` is an input)
                        first[0] = False
                    # (the call sites sit inside the samplers' torch.no_grad() regions: the piece is part of the differentiated graph)
                    with torch.cuda.stream(aux), torch.enable_grad():
                        piece()
                return run

            pieces = [on_aux(pc) for pc in self.sem_seg_head.training_pieces(sem_f, gt_sem_seg, sem_seg_losses)]
            # two pieces per call site at first (stride 4 and 8: the two long ones), then one: five pieces over four call sites
            F.defer_pieces([lambda a=pieces[0], b=pieces[1]: (a(), b())] + pieces[2:] if len(pieces) > 4 else pieces)
            for t in list(sem_f.values()) + [gt_sem_seg]:
                t.record_stream(aux)
            proposals, proposal_losses = self.proposal_generator(image_sizes, rpn_f, gt_instances)
            _, detector_losses = self.roi_heads(None, roi_f, proposals, gt_instances)
            F.flush_deferred()
        else:
            if aux is not None:
                main = torch.cuda.current_stream(self.device)
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    _, sem_seg_losses = self.sem_seg_head(sem_f, gt_sem_seg)
                for t in list(sem_f.values()) + [gt_sem_seg]:
                    t.record_stream(aux)
            else:
                _, sem_seg_losses = self.sem_seg_head(sem_f, gt_sem_seg)
            proposals, proposal_losses = self.proposal_generator(image_sizes, rpn_f, gt_instances)
            _, detector_losses = self.roi_heads(None, roi_f, proposals, gt_instances)
        F.join_aux_stream(self.device)  # semantic head and RPN losses
        losses = sem_seg_losses
        losses.update(proposal_losses)
        losses.update(detector_losses)
        return losses

    def inference(self, batched_inputs, do_postprocess=True):
        features, image_sizes, _ = self._backbone_features(batched_inputs)
        # as in training: the semantic head beside the detector branch, whose NMS / selection steps use a few CUs at a time
        aux = F.aux_stream(self.device) if next(iter(features.values())).is_cuda else None
        if aux is not None:
            main = torch.cuda.current_stream(self.device)
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                sem_seg_results, _ = self.sem_seg_head(features, None)
            for t in features.values():
                t.record_stream(aux)
        else:
            sem_seg_results, _ = self.sem_seg_head(features, None)
        proposals, _ = self.proposal_generator(image_sizes, features, None)
        detector_results, _ = self.roi_heads(None, features, proposals, None)
        if aux is not None:
            main.wait_stream(aux)
            sem_seg_results.record_stream(main)
            if getattr(sem_seg_results, "u2_argmax", None) is not None:
                sem_seg_results.u2_argmax.record_stream(main)
        if not do_postprocess:
            return detector_results, sem_seg_results
        out_sizes = [(inp.get("height", size[0]), inp.get("width", size[1])) for inp, size in zip(batched_inputs, image_sizes)]
        detector_rs = detector_postprocess_batch(detector_results, out_sizes)
        processed, sem_argmax = [], []
        fused_argmax = getattr(sem_seg_results, "u2_argmax", None)
        for i, (sem_seg_result, detector_r, image_size, (height, width)) in enumerate(zip(sem_seg_results, detector_rs, image_sizes,
                                                                                         out_sizes)):
            sem_seg_r = sem_seg_postprocess(sem_seg_result, image_size, height, width)
            processed.append({"sem_seg": sem_seg_r, "instances": detector_r})
            if fused_argmax is not None and (height, width) == tuple(image_size):
                sem_argmax.append(fused_argmax[i, : image_size[0], : image_size[1]])  # no second resampling: the kernel's argmax
            else:
                sem_argmax.append(sem_seg_r.argmax(dim=0))
        # the merge of all images: one launch, one host synchronisation (the reference loops and syncs per instance)
        mask_res = 2 * self.roi_heads.mask_pooler.output_size if getattr(self.roi_heads, "mask_on", False) else 0
        merged = combine_semantic_and_instance_outputs_batch(
            [p["instances"] for p in processed], sem_argmax, self.combine_overlap_thresh,
            self.combine_stuff_area_thresh, self.combine_instances_score_thresh, mask_res)
        for p, panoptic_r in zip(processed, merged):
            p["panoptic_seg"] = panoptic_r
        return processed


def build_model(cfg):
    """meta_arch/build.py:16-25."""
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
