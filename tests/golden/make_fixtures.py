"""Generates the golden fixtures under tests/golden/ by running the REFERENCE itself (imported from
/root/reference through small stand-ins for its missing third-party dependencies).  Runs only in the build
container (the reference does not travel to the GPU box); the resulting .npz / .json files are committed.

    python tests/golden/make_fixtures.py [--only NAME]

Without --only: kmeans, knn, ops, model_small, trajectory, inference.  The host-side fixtures need their stand-ins in place
before detectron2.data is imported and therefore run one per process:
    --only data | eval | pseudo_panoptic | label_prep | config

Stand-ins (none of them carries hot-path arithmetic except the two torchvision ops):
  * fvcore / iopath / yacs / omegaconf / termcolor / cv2 ... : registry, config node, weight init, smooth_l1 etc.
  * torchvision.ops.roi_align -> the reference's own vendored C++ op ROIAlignRotated (layers/csrc/ROIAlignRotated/
    ROIAlignRotated_cpu.cpp) compiled with torch.utils.cpp_extension and called at angle 0,
  * torchvision.ops.nms / batched_nms -> the rule stated in the reference's tests (tests/layers/test_nms_rotated.py:44-66).
"""
import argparse
import copy
import glob
import importlib.abc
import importlib.machinery
import importlib.util
import json
import math
import os
import sys
import types
import zlib
from unittest import mock

import numpy as np
import torch
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


# ------------------------------------------------------------------------------------------------
# third-party stand-ins
# ------------------------------------------------------------------------------------------------
class _AutoMod(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = mock.MagicMock(name="%s.%s" % (self.__name__, name))
        setattr(self, name, m)
        return m


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("fvcore", "iopath", "yacs", "omegaconf", "hydra", "torchvision", "pycocotools", "termcolor", "cv2",
             "panopticapi", "tensorboard", "lvis", "shapely", "timm", "fairscale", "black", "pykeops", "clip", "faiss")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = _AutoMod(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_standins():
    sys.meta_path.insert(0, _Finder())
    from torch import nn

    import fvcore.common.registry as R

    class Registry:
        def __init__(self, name):
            self._name, self._obj_map = name, {}

        def _do_register(self, name, obj):
            assert name not in self._obj_map, name
            self._obj_map[name] = obj

        def register(self, obj=None):
            if obj is None:
                def deco(f):
                    self._do_register(f.__name__, f)
                    return f
                return deco
            self._do_register(obj.__name__, obj)

        def get(self, name):
            return self._obj_map[name]

    R.Registry = Registry

    import fvcore.common.config as C

    class CfgNode(dict):
        def __init__(self, init=None, **kw):
            super().__init__()
            for k, v in (init or {}).items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def clone(self):
            return copy.deepcopy(self)

        def freeze(self):
            pass

        def defrost(self):
            pass

        def is_frozen(self):
            return False

        def merge_from_other(self, o):
            for k, v in o.items():
                if isinstance(v, dict) and k in self and isinstance(self[k], dict):
                    self[k].merge_from_other(v)
                else:
                    self[k] = CfgNode(v) if isinstance(v, dict) else v

        @classmethod
        def load_yaml_with_base(cls, fn, allow_unsafe=False):
            d = yaml.unsafe_load(open(fn))

            def fix(x):
                if isinstance(x, dict):
                    return {k: fix(v) for k, v in x.items()}
                if isinstance(x, str) and x.startswith("("):
                    return eval(x)
                return x

            d = fix(d)
            base = d.pop("_BASE_", None)
            if base:
                b = cls.load_yaml_with_base(os.path.join(os.path.dirname(fn), base))

                def mrg(a, b_):
                    for k, v in a.items():
                        if isinstance(v, dict) and isinstance(b_.get(k), dict):
                            mrg(v, b_[k])
                        else:
                            b_[k] = v

                mrg(d, b)
                return b
            return d

        def merge_from_other_cfg(self, o):
            self.merge_from_other(o)

        def merge_from_list(self, l):
            for k, v in zip(l[0::2], l[1::2]):
                d = self
                ks = k.split(".")
                for kk in ks[:-1]:
                    d = d[kk]
                d[ks[-1]] = v

        def dump(self, **kw):
            return yaml.safe_dump(json.loads(json.dumps(self)))

    C.CfgNode = CfgNode

    import fvcore.nn.weight_init as W

    def c2_xavier_fill(m):
        nn.init.kaiming_uniform_(m.weight, a=1)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    def c2_msra_fill(m):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)

    W.c2_xavier_fill, W.c2_msra_fill = c2_xavier_fill, c2_msra_fill

    import fvcore.nn as FN

    def smooth_l1_loss(input, target, beta, reduction="none"):
        if beta < 1e-5:
            loss = torch.abs(input - target)
        else:
            n = torch.abs(input - target)
            loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
        if reduction == "mean":
            return loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
        if reduction == "sum":
            return loss.sum()
        return loss

    FN.smooth_l1_loss = smooth_l1_loss

    import fvcore.common.history_buffer as HB

    class HistoryBuffer:
        def __init__(self, max_length=1000000):
            self._d = []

        def update(self, v, it=None):
            self._d.append((v, it))

        def latest(self):
            return self._d[-1][0]

        def values(self):
            return self._d

    HB.HistoryBuffer = HistoryBuffer

    import iopath.common.file_io as IO

    class PathManagerBase:
        def register_handler(self, *a, **k):
            pass

        def open(self, p, mode="r", **k):
            return open(p, mode)

        def isfile(self, p):
            return os.path.isfile(p)

        def exists(self, p):
            return os.path.exists(p)

        def get_local_path(self, p, **k):
            return p

        def ls(self, p):
            return os.listdir(p)

        def mkdirs(self, p):
            os.makedirs(p, exist_ok=True)

    class PathHandler:
        pass

    IO.PathManager, IO.PathHandler = PathManagerBase, PathHandler
    IO.HTTPURLHandler = IO.OneDrivePathHandler = PathHandler

    import fvcore.transforms.transform as FT

    class Transform:
        """fvcore.transforms.transform.Transform restated: attribute capture, box = bounding box of the mapped corners,
        label maps and polygons fall back to the image / coordinate maps."""

        def _set_attributes(self, params=None):
            if params:
                for k, v in params.items():
                    if k != "self" and not k.startswith("_"):
                        setattr(self, k, v)

        @classmethod
        def register_type(cls, *a, **k):
            return lambda f: f

        def apply_segmentation(self, segmentation):
            return self.apply_image(segmentation)

        def apply_box(self, box):
            idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
            coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
            coords = self.apply_coords(coords).reshape((-1, 4, 2))
            return np.concatenate((coords.min(axis=1), coords.max(axis=1)), axis=1)

        def apply_polygons(self, polygons):
            return [self.apply_coords(p) for p in polygons]

    class NoOpTransform(Transform):
        def apply_image(self, img):
            return img

        def apply_coords(self, coords):
            return coords

    class HFlipTransform(Transform):
        def __init__(self, width):
            self.width = width

        def apply_image(self, img):
            return np.flip(img, axis=1)

        def apply_coords(self, coords):
            coords[:, 0] = self.width - coords[:, 0]
            return coords

    class VFlipTransform(Transform):
        def __init__(self, height):
            self.height = height

        def apply_image(self, img):
            return np.flip(img, axis=0)

        def apply_coords(self, coords):
            coords[:, 1] = self.height - coords[:, 1]
            return coords

    class TransformList(Transform):
        def __init__(self, transforms):
            flat = []
            for t in transforms:
                flat.extend(t.transforms if isinstance(t, TransformList) else [t])
            self.transforms = [t for t in flat if not isinstance(t, NoOpTransform)]

        def __getattr__(self, name):
            if name.startswith("apply_"):
                def chain(x):
                    for t in self.transforms:
                        x = getattr(t, name)(x)
                    return x
                return chain
            raise AttributeError(name)

        def __len__(self):
            return len(self.transforms)

    # the chained forms must win over the per-transform defaults inherited from Transform
    for _n in ("apply_segmentation", "apply_box", "apply_polygons"):
        setattr(TransformList, _n, (lambda n: lambda self, x: TransformList.__getattr__(self, n)(x))(_n))

    real = {"Transform": Transform, "TransformList": TransformList, "HFlipTransform": HFlipTransform,
            "VFlipTransform": VFlipTransform, "NoOpTransform": NoOpTransform}
    names = ["Transform", "TransformList", "BlendTransform", "CropTransform", "PadTransform", "GridSampleTransform",
             "HFlipTransform", "VFlipTransform", "NoOpTransform", "ScaleTransform"]
    for n in names:
        setattr(FT, n, real.get(n) or type(n, (Transform,), {}))
    FT.__all__ = names
    import fvcore.transforms as FTT

    FTT.HFlipTransform, FTT.NoOpTransform = FT.HFlipTransform, FT.NoOpTransform

    import fvcore.common.checkpoint as CK

    class Checkpointer:
        def __init__(self, model, save_dir="", **kw):
            self.model = model

    CK.Checkpointer = Checkpointer
    CK.PeriodicCheckpointer = type("PeriodicCheckpointer", (), {})
    import fvcore.common.param_scheduler as PS

    class ParamScheduler:
        pass

    for n in ["ParamScheduler", "CosineParamScheduler", "MultiStepParamScheduler", "StepWithFixedGammaParamScheduler",
              "CompositeParamScheduler", "ConstantParamScheduler", "LinearParamScheduler"]:
        setattr(PS, n, type(n, (ParamScheduler,), {}))

    import torchvision
    import torchvision.ops as OPS
    import torchvision.ops.boxes as BOX

    torchvision.__version__ = "0.25.0"

    def nms(boxes, scores, thr):
        """torchvision rule as stated in the reference tests (test_nms_rotated.py:44-66): survivors have iou <= thr."""
        if boxes.numel() == 0:
            return torch.empty((0,), dtype=torch.int64)
        order = scores.argsort(descending=True, stable=True)
        b = boxes[order]
        n = len(b)
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        keep, sup = [], torch.zeros(n, dtype=torch.bool)
        for i in range(n):
            if sup[i]:
                continue
            keep.append(i)
            xx1, yy1 = torch.maximum(b[i, 0], b[i + 1:, 0]), torch.maximum(b[i, 1], b[i + 1:, 1])
            xx2, yy2 = torch.minimum(b[i, 2], b[i + 1:, 2]), torch.minimum(b[i, 3], b[i + 1:, 3])
            inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
            sup[i + 1:] |= inter / (area[i] + area[i + 1:] - inter) > thr
        return order[torch.tensor(keep, dtype=torch.int64)]

    def batched_nms(boxes, scores, idxs, thr):
        """per-group NMS on un-offset boxes (torchvision's _batched_nms_vanilla), merged by descending score."""
        if boxes.numel() == 0:
            return torch.empty((0,), dtype=torch.int64)
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for cid in torch.unique(idxs):
            cur = torch.where(idxs == cid)[0]
            keep_mask[cur[nms(boxes[cur], scores[cur], thr)]] = True
        kept = torch.where(keep_mask)[0]
        return kept[scores[kept].argsort(descending=True, stable=True)]

    def box_iou(b1, b2):
        """torchvision.ops.box_iou: inter / (area1 + area2 - inter)."""
        a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
        a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
        wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        return inter / (a1[:, None] + a2 - inter)

    OPS.nms, BOX.batched_nms, BOX.nms, OPS.boxes, OPS.box_iou, BOX.box_iou = nms, batched_nms, nms, BOX, box_iou, box_iou

    class RoIPool(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    OPS.RoIPool = RoIPool
    OPS.roi_align = None  # patched after detectron2._C is built
    import termcolor

    termcolor.colored = lambda s, *a, **k: s
    import cv2
    import fvcore
    import iopath

    cv2.__version__, fvcore.__version__, iopath.__version__ = "4.8.0", "0.1.5", "0.1.9"
    import omegaconf

    omegaconf.DictConfig = type("DictConfig", (dict,), {})
    omegaconf.ListConfig = type("ListConfig", (list,), {})


def import_reference():
    """Imports detectron2 from /root/reference with roi_align served by the vendored C++ ROIAlignRotated at 0 deg."""
    install_standins()
    sys.path.insert(0, REF)
    from torch.utils.cpp_extension import load

    csrc = os.path.join(REF, "detectron2", "layers", "csrc")
    srcs = [os.path.join(csrc, "vision.cpp")] + [s for s in glob.glob(os.path.join(csrc, "**", "*.cpp"), recursive=True)
                                                  if not s.endswith("vision.cpp")]
    ext = load(name="_C", sources=srcs, extra_include_paths=[csrc], build_directory=_build_dir(), verbose=False)
    sys.modules["detectron2._C"] = ext
    import detectron2  # noqa: F401
    from detectron2.layers.roi_align_rotated import roi_align_rotated
    RA = sys.modules["detectron2.layers.roi_align"]

    def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
        assert aligned, "only the aligned=True form is on the U2Seg path"
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        # (b, x0, y0, x1, y1) -> (b, cx, cy, w, h, 0): the rotated op centres at (cx, cy)*scale - 0.5 like aligned ROIAlign
        r = torch.stack([rois[:, 0], (rois[:, 1] + rois[:, 3]) / 2, (rois[:, 2] + rois[:, 4]) / 2, rois[:, 3] - rois[:, 1],
                         rois[:, 4] - rois[:, 2], torch.zeros_like(rois[:, 0])], dim=1)
        if input.dtype != torch.float32:
            # torchvision registers roi_align for autocast with a wrapper that computes in fp32 and returns the input's dtype
            # (torchvision/csrc/ops/autocast/roi_align_kernel.cpp); the vendored rotated op has no bf16 kernel either
            return roi_align_rotated(input.float(), r.float(), output_size, spatial_scale, max(sampling_ratio, 0)).to(input.dtype)
        return roi_align_rotated(input, r, output_size, spatial_scale, max(sampling_ratio, 0))

    RA.roi_align = roi_align
    return roi_align


def _build_dir():
    d = "/tmp/u2seg_ref_ext"
    os.makedirs(d, exist_ok=True)
    return d


# ------------------------------------------------------------------------------------------------
def det_fill(name, tensor):
    """Deterministic, name-keyed parameter fill shared by the fixture generator and the tests."""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    if name.endswith("running_var"):
        return 0.5 + torch.rand(tensor.shape, generator=g)
    if name.endswith("running_mean"):
        return 0.1 * torch.randn(tensor.shape, generator=g)
    if name.endswith("num_batches_tracked"):
        return torch.zeros_like(tensor)
    if tensor.dim() == 1:
        if name.endswith(".bias"):
            return 0.05 * torch.randn(tensor.shape, generator=g)
        return 1.0 + 0.1 * torch.randn(tensor.shape, generator=g)  # norm weights
    fan_in = tensor[0].numel()
    return torch.randn(tensor.shape, generator=g) * (1.0 / math.sqrt(fan_in))


def to_ref_batch(batch):
    from detectron2.structures import BitMasks, Boxes, Instances

    out = []
    for x in batch:
        inst = Instances(x["instances"].image_size)
        inst.gt_boxes = Boxes(x["instances"].gt_boxes.tensor.clone())
        inst.gt_classes = x["instances"].gt_classes.clone()
        inst.gt_masks = BitMasks(x["instances"].gt_masks.tensor.clone())
        out.append({"image": x["image"], "instances": inst, "sem_seg": x["sem_seg"], "height": x["height"], "width": x["width"]})
    return out


def gen_model_fixture(name, hw, nimg, seed):
    os.environ.setdefault("CLUSTER_NUM", "800")
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.utils.events import EventStorage

    from u2seg_amd.data import make_synthetic_batch

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.WEIGHTS = ""
    model = build_model(cfg)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(det_fill(k, v))
    model.train()
    batch = to_ref_batch(make_synthetic_batch(nimg, height=hw[0], width=hw[1]))
    torch.manual_seed(seed)
    with EventStorage():
        losses = model(batch)
        total = sum(losses.values())
        total.backward()
    grads = {}
    for k, p in model.named_parameters():
        if p.grad is not None:
            grads[k] = float(p.grad.double().norm())
    picked = ["backbone.bottom_up.stem.conv1.weight", "backbone.bottom_up.res2.0.conv1.weight",
              "backbone.bottom_up.res3.0.shortcut.weight", "backbone.bottom_up.res4.5.conv3.norm.weight",
              "backbone.bottom_up.res5.2.conv2.weight", "backbone.fpn_lateral3.weight", "backbone.fpn_output2.weight",
              "proposal_generator.rpn_head.conv.weight", "proposal_generator.rpn_head.anchor_deltas.bias",
              "roi_heads.box_head.0.fc1.weight", "roi_heads.box_head.2.fc2.bias", "roi_heads.box_predictor.1.cls_score.weight",
              "roi_heads.box_predictor.0.bbox_pred.weight", "roi_heads.mask_head.mask_fcn1.weight",
              "roi_heads.mask_head.deconv.weight", "roi_heads.mask_head.predictor.weight", "sem_seg_head.p2.0.weight",
              "sem_seg_head.p5.4.norm.weight", "sem_seg_head.predictor.weight"]
    out = {"config": "u2seg_R50_800.yaml", "image_hw": list(hw), "num_images": nimg, "seed": seed,
           "weights": "tests/golden/make_fixtures.py:det_fill", "data": "u2seg_amd.data.make_synthetic_batch(start_index=0)",
           "losses": {k: float(v) for k, v in losses.items()}, "num_params": sum(p.numel() for p in model.parameters()),
           "grad_norms": {k: grads[k] for k in picked}, "num_state_entries": len(sd),
           "state_dict_keys_crc32": zlib.crc32("\n".join(sd.keys()).encode())}
    json.dump(out, open(os.path.join(HERE, name + ".json"), "w"), indent=1)
    print("wrote", name, out["losses"])


TRAJ_PARAMS = ["backbone.bottom_up.stem.conv1.weight", "backbone.bottom_up.res4.5.conv3.norm.weight",
               "backbone.fpn_output2.weight", "roi_heads.box_predictor.1.cls_score.weight",
               "roi_heads.mask_head.deconv.weight", "sem_seg_head.predictor.bias"]
TRAJ_OVERRIDES = ["SOLVER.WARMUP_ITERS", 2, "SOLVER.WARMUP_FACTOR", 0.25, "SOLVER.STEPS", (3,), "SOLVER.MAX_ITER", 4,
                  "SOLVER.WEIGHT_DECAY_NORM", 0.001]


def gen_trajectory_fixture(name, hw, nimg, seed, steps=4):
    """Four SGD steps of the reference: its PanopticFPN, its build_optimizer (per-parameter L2 clip wrapped around
    torch.optim.SGD, solver/build.py:29-153) and its plain-python WarmupMultiStepLR (solver/lr_scheduler.py:141-173;
    equal to the fvcore composite build_lr_scheduler assembles whenever no milestone falls inside the warm-up), fp32 on
    the CPU.  The schedule is compressed (warm-up over 2 iterations from 0.25, one milestone at 3) so that warm-up, the
    plateau and a decay are all inside four steps; a fresh batch (start_index = step * nimg) per step."""
    os.environ.setdefault("CLUSTER_NUM", "800")
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.solver import build_optimizer
    from detectron2.solver.lr_scheduler import WarmupMultiStepLR
    from detectron2.utils.events import EventStorage

    from u2seg_amd.data import make_synthetic_batch

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.WEIGHTS = ""
    cfg.merge_from_list(list(TRAJ_OVERRIDES))
    model = build_model(cfg)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(det_fill(k, v))
    model.train()
    opt = build_optimizer(cfg, model)
    sched = WarmupMultiStepLR(opt, list(cfg.SOLVER.STEPS), cfg.SOLVER.GAMMA, cfg.SOLVER.WARMUP_FACTOR,
                              cfg.SOLVER.WARMUP_ITERS, cfg.SOLVER.WARMUP_METHOD)
    init = {k: dict(model.named_parameters())[k].detach().clone() for k in TRAJ_PARAMS}
    torch.manual_seed(seed)
    per_step, lrs = [], []
    with EventStorage() as storage:
        for it in range(steps):
            batch = to_ref_batch(make_synthetic_batch(nimg, height=hw[0], width=hw[1], start_index=it * nimg))
            losses = model(batch)
            opt.zero_grad()
            sum(losses.values()).backward()
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
            storage.step()
            per_step.append({k: float(v) for k, v in losses.items()})
            print("step", it, "lr", lrs[-1], "total", sum(per_step[-1].values()))
    params = dict(model.named_parameters())
    buffers = dict(model.named_buffers())
    out = {"config": "u2seg_R50_800.yaml", "overrides": [list(x) if isinstance(x, tuple) else x for x in TRAJ_OVERRIDES],
           "image_hw": list(hw), "num_images": nimg, "seed": seed, "steps": steps, "lr": lrs, "losses": per_step,
           "param_norm": {k: float(params[k].double().norm()) for k in TRAJ_PARAMS},
           "param_sum": {k: float(params[k].double().sum()) for k in TRAJ_PARAMS},
           "param_delta_norm": {k: float((params[k].detach() - init[k]).double().norm()) for k in TRAJ_PARAMS},
           "running_mean_norm": {k: float(buffers[k].double().norm()) for k in
                                 ["backbone.bottom_up.stem.conv1.norm.running_mean",
                                  "backbone.bottom_up.res5.2.conv3.norm.running_var"]},
           "num_batches_tracked": int(buffers["backbone.bottom_up.stem.conv1.norm.num_batches_tracked"])}
    json.dump(out, open(os.path.join(HERE, name + ".json"), "w"), indent=1)
    print("wrote", name)


def gen_inference_fixture(name, hw, nimg):
    """Reference PanopticFPN.inference on synthetic images with name-keyed weights (eval-mode BN)."""
    os.environ.setdefault("CLUSTER_NUM", "800")
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model

    from u2seg_amd.data import make_synthetic_batch

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.WEIGHTS = ""
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 0.0015  # random weights: ~uniform 1/801 scores, keep a useful number of boxes
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v))
    model.eval()
    if isinstance(hw[0], int):
        batch = to_ref_batch(make_synthetic_batch(nimg, height=hw[0], width=hw[1]))
    else:  # ragged batch: one size per image, results requested at 1.5x the input resolution (postprocessing.py:9-100)
        batch = []
        for i, (h, w) in enumerate(hw):
            x = to_ref_batch(make_synthetic_batch(1, start_index=i, height=h, width=w))[0]
            x["height"], x["width"] = int(1.5 * h), int(1.5 * w)
            batch.append(x)
    with torch.no_grad():
        out = model([{k: v for k, v in x.items() if k != "instances"} for x in batch])
    arrays = {"score_thresh": np.array(0.0015), "sizes": np.array([x["image"].shape[1:] for x in batch]),
              "out_sizes": np.array([[x["height"], x["width"]] for x in batch])}
    for i, o in enumerate(out):
        inst = o["instances"]
        arrays["boxes_%d" % i] = inst.pred_boxes.tensor.numpy()
        arrays["scores_%d" % i] = inst.scores.numpy()
        arrays["classes_%d" % i] = inst.pred_classes.numpy()
        arrays["mask_areas_%d" % i] = inst.pred_masks.flatten(1).sum(1).numpy()
        arrays["sem_argmax_%d" % i] = o["sem_seg"].argmax(0).to(torch.uint8).numpy()
        arrays["sem_logit_absmean_%d" % i] = np.array(float(o["sem_seg"].abs().mean()))
        arrays["panoptic_%d" % i] = o["panoptic_seg"][0].numpy().astype(np.int16)
        arrays["panoptic_info_%d" % i] = np.array(json.dumps(o["panoptic_seg"][1]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print("wrote", name, [len(o["instances"]) for o in out], [len(o["panoptic_seg"][1]) for o in out])


def gen_op_fixtures(roi_align_ref):
    g = torch.Generator().manual_seed(7)
    out = {}
    # --- ROIAlign fwd/bwd from the vendored C++ op ------------------------------------------------
    feat = torch.randn((2, 6, 20, 24), generator=g)
    rois = torch.tensor([[0, 4.0, 4.0, 60.0, 50.0], [1, -10.0, -8.0, 30.0, 20.0], [0, 0.0, 0.0, 96.0, 80.0],
                         [1, 50.2, 33.7, 50.9, 34.1], [0, 90.0, 70.0, 120.0, 100.0], [1, 10.0, 10.0, 10.0, 10.0],
                         [0, 13.3, 7.7, 77.1, 41.9], [1, 200.0, 200.0, 260.0, 240.0]])
    for ps, tag in ((7, "p7"), (14, "p14")):
        f = feat.clone().requires_grad_(True)
        y = roi_align_ref(f, rois, ps, 0.25, 0, True)
        w = torch.randn(y.shape, generator=g)
        (y * w).sum().backward()
        out["roi_%s_out" % tag], out["roi_%s_w" % tag], out["roi_%s_grad" % tag] = y.detach().numpy(), w.numpy(), f.grad.numpy()
    out["roi_feat"], out["roi_rois"] = feat.numpy(), rois.numpy()
    # reference unit-test known answer (tests/layers/test_roi_align.py:14-47)
    ramp = torch.arange(25, dtype=torch.float32).reshape(1, 1, 5, 5)
    out["roi_ramp_out"] = roi_align_ref(ramp, torch.tensor([[0, 1.0, 1.0, 3.0, 3.0]]), 4, 1.0, 0, True).numpy()
    out["roi_ramp_expected"] = np.array([[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0],
                                         [12.0, 12.5, 13.0, 13.5]], dtype=np.float32)
    # mask crop (BitMasks.crop_and_resize, structures/masks.py:191-218)
    from detectron2.structures import BitMasks

    masks = torch.rand((5, 40, 52), generator=g) > 0.45
    mboxes = torch.tensor([[2.0, 3.0, 30.0, 33.0], [0.0, 0.0, 52.0, 40.0], [10.5, 7.25, 20.75, 30.5], [40.0, 30.0, 60.0, 45.0],
                           [5.0, 5.0, 6.0, 6.5]])
    out["crop_masks"], out["crop_boxes"] = masks.numpy(), mboxes.numpy()
    out["crop_out"] = BitMasks(masks).crop_and_resize(mboxes, 28).numpy()
    # --- Matcher ----------------------------------------------------------------------------------
    from detectron2.modeling.matcher import Matcher
    from detectron2.structures import Boxes, pairwise_iou

    gtb = torch.tensor([[10.0, 10, 60, 70], [30, 20, 100, 90], [100, 100, 140, 130], [0, 0, 20, 20]])
    cand = torch.rand((300, 2), generator=g) * 120
    cand = torch.cat([cand, cand + 5 + torch.rand((300, 2), generator=g) * 60], dim=1)
    cand = torch.cat([cand, gtb[:2]], dim=0)
    iou = pairwise_iou(Boxes(gtb), Boxes(cand))
    m1 = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)(iou)
    m2 = Matcher([0.5], [0, 1], allow_low_quality_matches=False)(iou)
    out.update(match_gt=gtb.numpy(), match_cand=cand.numpy(), match_iou=iou.numpy(), match_rpn_idx=m1[0].numpy(),
               match_rpn_lab=m1[1].numpy(), match_roi_idx=m2[0].numpy(), match_roi_lab=m2[1].numpy())
    # known answer of tests/modeling/test_matcher.py:12-25
    kq = torch.tensor([[0.15, 0.45, 0.2, 0.6], [0.3, 0.65, 0.05, 0.1], [0.05, 0.4, 0.25, 0.4]])
    km = Matcher([0.3, 0.5], [0, -1, 1], allow_low_quality_matches=True)(kq)
    out.update(match_known_q=kq.numpy(), match_known_idx=km[0].numpy(), match_known_lab=km[1].numpy())
    # --- anchors ----------------------------------------------------------------------------------
    from detectron2.config import get_cfg
    from detectron2.layers import ShapeSpec
    from detectron2.modeling.anchor_generator import DefaultAnchorGenerator

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    ag = DefaultAnchorGenerator(cfg, [ShapeSpec(stride=s) for s in (4, 8, 16, 32, 64)])
    grids = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    anc = ag([torch.zeros(1, 1, h, w) for h, w in grids])
    for i, a in enumerate(anc):
        out["anchors_l%d" % i] = a.tensor.numpy()
    # --- box coding, clip, levels ------------------------------------------------------------------
    from detectron2.modeling.box_regression import Box2BoxTransform
    from detectron2.modeling.poolers import assign_boxes_to_levels

    src = cand[:64]
    tgt = src + torch.randn((64, 4), generator=g) * 4
    tgt[:, 2:] = torch.maximum(tgt[:, 2:], tgt[:, :2] + 1)
    for wts, tag in (((1.0, 1.0, 1.0, 1.0), "rpn"), ((10.0, 10.0, 5.0, 5.0), "s0"), ((30.0, 30.0, 15.0, 15.0), "s2")):
        t = Box2BoxTransform(weights=wts)
        d = t.get_deltas(src, tgt)
        dn = d + torch.randn(d.shape, generator=g) * 0.5
        dn[0, 2] = 40.0  # exercises the log(1000/16) clamp
        out["b2b_%s_deltas" % tag], out["b2b_%s_noisy" % tag] = d.numpy(), dn.numpy()
        out["b2b_%s_applied" % tag] = t.apply_deltas(dn, src).numpy()
    out["b2b_src"], out["b2b_tgt"] = src.numpy(), tgt.numpy()
    lv_boxes = torch.cat([cand[:200], torch.tensor([[0, 0, 112.0, 112.0], [0, 0, 224.0, 224.0], [0, 0, 448, 448], [0, 0, 896, 896.0],
                                                     [0, 0, 111.99, 112.0], [0, 0, 1333, 800], [5, 5, 5, 5]])])
    out["lvl_boxes"] = lv_boxes.numpy()
    out["lvl_out"] = assign_boxes_to_levels([Boxes(lv_boxes)], 2, 5, 224, 4).numpy()
    # --- subsample_labels (seeded CPU randperm) ----------------------------------------------------
    from detectron2.modeling.sampling import subsample_labels

    labels = torch.randint(-1, 3, (500,), generator=g)
    torch.manual_seed(11)
    pos, neg = subsample_labels(labels, 64, 0.25, 2)
    out.update(sub_labels=labels.numpy(), sub_pos=pos.numpy(), sub_neg=neg.numpy())
    # --- NMS by the reference tests' python rule ---------------------------------------------------
    spec = importlib.util.spec_from_file_location("ref_test_nms_rotated", os.path.join(REF, "tests/layers/test_nms_rotated.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    nb = torch.rand((400, 2), generator=g) * 200
    nb = torch.cat([nb, nb + 10 + torch.rand((400, 2), generator=g) * 80], dim=1)
    ns = torch.rand(400, generator=g)
    out.update(nms_boxes=nb.numpy(), nms_scores=ns.numpy())
    for thr in (0.5, 0.65):
        out["nms_keep_%02d" % int(thr * 100)] = mod.TestNMSRotated().reference_horizontal_nms(nb, ns, thr).numpy()
    # --- panoptic merge ---------------------------------------------------------------------------
    from detectron2.modeling.meta_arch.panoptic_fpn import combine_semantic_and_instance_outputs
    from detectron2.structures import Instances

    H, W = 96, 128
    sem = torch.randint(0, 6, (H // 16, W // 16), generator=g).repeat_interleave(16, 0).repeat_interleave(16, 1)
    pm = torch.zeros((6, H, W), dtype=torch.bool)
    rects = [(5, 5, 60, 70), (30, 40, 90, 120), (0, 0, 20, 20), (50, 60, 95, 127), (10, 10, 58, 68), (70, 5, 90, 30)]
    for i, (y0, x0, y1, x1) in enumerate(rects):
        pm[i, y0:y1, x0:x1] = True
    inst = Instances((H, W))
    inst.pred_masks, inst.scores = pm, torch.tensor([0.9, 0.8, 0.3, 0.75, 0.95, 0.6])
    inst.pred_classes = torch.tensor([3, 7, 1, 2, 9, 4])
    pan, info = combine_semantic_and_instance_outputs(inst, sem, 0.5, 4096 // 8, 0.5)
    out.update(pan_sem=sem.numpy(), pan_masks=pm.numpy(), pan_scores=inst.scores.numpy(), pan_classes=inst.pred_classes.numpy(),
               pan_out=pan.numpy(), pan_info=np.array(json.dumps(info)))
    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), **out)
    print("wrote ops_golden.npz with", len(out), "arrays")


DATA_ROOT = os.path.join(HERE, "data_small")
DATA_INPUT_OPTS = ["INPUT.MIN_SIZE_TRAIN", (48, 64, 80), "INPUT.MAX_SIZE_TRAIN", 100, "INPUT.MIN_SIZE_TEST", 64,
                   "INPUT.MAX_SIZE_TEST", 100]


def make_small_dataset():
    """Writes tests/golden/data_small/: 4 jpg images, their semantic label maps and a COCO instances json at the paths
    the builtin registration expects for CLUSTER_NUM=800 (builtin.py:59-118).  Masks are RLE (compressed strings, one
    uncompressed list); one annotation is a crowd region, one image has only a crowd annotation."""
    from PIL import Image

    from u2seg_amd.data import rle

    rs = np.random.RandomState(0)
    img_dir = os.path.join(DATA_ROOT, "coco", "train2017")
    ann_dir = os.path.join(DATA_ROOT, "prepare_ours", "u2seg_annotations", "ins_annotations")
    sem_dir = os.path.join(DATA_ROOT, "prepare_ours", "u2seg_annotations", "panoptic_annotations",
                           "panoptic_stuff_cocotrain_800")
    for d in (img_dir, ann_dir, sem_dir):
        os.makedirs(d, exist_ok=True)
    sizes = [(48, 64), (60, 40), (50, 50), (36, 72)]  # (h, w): landscape, portrait, square, wide
    images, annotations, aid = [], [], 1
    for i, (h, w) in enumerate(sizes):
        name = "%012d" % (i + 1)
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([xx * 255.0 / w, yy * 255.0 / h, (xx + yy) * 255.0 / (h + w)], axis=2)
        img = np.clip(base + rs.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(img_dir, name + ".jpg"), quality=92)
        sem = (rs.randint(0, 28, (h // 8 + 1, w // 8 + 1)).repeat(8, 0).repeat(8, 1)[:h, :w]).astype(np.uint8)
        sem[rs.rand(h, w) < 0.04] = 255
        Image.fromarray(sem, mode="L").save(os.path.join(sem_dir, name + ".png"))
        images.append({"id": 10 * (i + 1), "file_name": name + ".jpg", "height": h, "width": w})
        n_inst = [3, 2, 1, 2][i]
        for k in range(n_inst):
            bw, bh = rs.uniform(0.3, 0.7) * w, rs.uniform(0.3, 0.7) * h
            x0, y0 = rs.uniform(0, w - bw), rs.uniform(0, h - bh)
            m = (((xx + 0.5 - (x0 + bw / 2)) / (bw / 2)) ** 2 + ((yy + 0.5 - (y0 + bh / 2)) / (bh / 2)) ** 2 <= 1).astype(np.uint8)
            segm = rle.encode(m)
            if i == 0 and k == 1:  # one uncompressed RLE (counts as a list), coco.py:189-191
                segm = {"size": segm["size"], "counts": rle.counts_of(segm)}
            crowd = 1 if (i == 2 or (i == 0 and k == 2)) else 0  # image 3 holds nothing but a crowd region
            annotations.append({"id": aid, "image_id": 10 * (i + 1), "category_id": [5, 17, 800, 333][(i + k) % 4],
                                "iscrowd": crowd, "bbox": [round(x0, 2), round(y0, 2), round(bw, 2), round(bh, 2)],
                                "area": float(m.sum()), "segmentation": segm})
            aid += 1
    cats = [{"id": c + 1, "name": str(c + 1), "supercategory": str(c + 1)} for c in range(800)]
    json.dump({"images": images, "annotations": annotations, "categories": cats},
              open(os.path.join(ann_dir, "cocotrain_800.json"), "w"))
    print("wrote", DATA_ROOT)


def install_data_standins():
    """pycocotools (absent) for the reference's data path: the json index is plain bookkeeping; mask decode / compress go
    through u2seg_amd.data.rle, which tests/test_host_logic.py pins on the RLE strings the reference's own tests carry."""
    from u2seg_amd.data import rle

    import pycocotools.coco as PC
    import pycocotools.mask as PM

    class COCO:
        def __init__(self, annotation_file):
            from collections import defaultdict

            data = json.load(open(annotation_file))
            self.dataset = data
            self.imgs = {im["id"]: im for im in data["images"]}
            self.anns = {a["id"]: a for a in data["annotations"]}
            self.cats = {c["id"]: c for c in data["categories"]}
            self.imgToAnns = defaultdict(list)
            for a in data["annotations"]:
                self.imgToAnns[a["image_id"]].append(a)

        def getCatIds(self):
            return list(self.cats.keys())

        def loadCats(self, ids):
            return [self.cats[i] for i in ids]

        def loadImgs(self, ids):
            return [self.imgs[i] for i in ids]

    PC.COCO = COCO
    PM.decode = rle.decode
    PM.encode = rle.encode
    PM.area = rle.area
    PM.toBbox = rle.to_bbox
    PM.frPyObjects = lambda obj, h, w: rle.compress(obj)
    import fvcore.common.timer as TM

    class Timer:
        def seconds(self):
            return 0.0

    TM.Timer = Timer


def gen_data_fixture():
    """Runs the reference's registration (builtin.py with DETECTRON2_DATASETS pointing at data_small), dataset-dict
    loading, DatasetMapper (train and test) and samplers on the small dataset and records their outputs."""
    import itertools

    import tempfile

    os.environ.setdefault("CLUSTER_NUM", "800")
    if not os.path.exists(os.path.join(DATA_ROOT, "coco")):
        make_small_dataset()
    # the fork hard-wires the dataset root to ./datasets (builtin.py:279): run from a directory where that is data_small
    work = tempfile.mkdtemp()
    os.symlink(DATA_ROOT, os.path.join(work, "datasets"))
    os.chdir(work)
    install_standins()
    install_data_standins()
    sys.path.insert(0, REF)
    import types as _types

    sys.modules["detectron2._C"] = _types.ModuleType("detectron2._C")  # the data path touches no compiled op
    from detectron2.config import get_cfg
    from detectron2.data import DatasetCatalog, DatasetMapper, MetadataCatalog
    from detectron2.data.build import get_detection_dataset_dicts
    from detectron2.data.samplers import InferenceSampler, TrainingSampler

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    cfg.merge_from_list(list(DATA_INPUT_OPTS))
    name = cfg.DATASETS.TRAIN[0]
    raw = DatasetCatalog.get(name)
    dicts = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)

    def rel(d):
        d = copy.deepcopy(d)
        for k in ("file_name", "sem_seg_file_name"):
            if k in d:
                d[k] = os.path.relpath(os.path.realpath(d[k]), os.path.realpath(DATA_ROOT))
        for a in d.get("annotations", []):
            a["bbox_mode"] = int(a["bbox_mode"])
            if isinstance(a.get("segmentation"), dict) and isinstance(a["segmentation"]["counts"], bytes):
                a["segmentation"]["counts"] = a["segmentation"]["counts"].decode()
        return d

    meta = MetadataCatalog.get(name)
    listing = {"train_name": name, "test_name": cfg.DATASETS.TEST[0], "num_raw": len(raw),
               "dicts": [rel(d) for d in dicts],
               "meta": {"evaluator_type": meta.evaluator_type, "ignore_label": meta.ignore_label,
                        "num_thing_classes": len(meta.thing_classes), "num_stuff_classes": len(meta.stuff_classes),
                        "thing_id_800": meta.thing_dataset_id_to_contiguous_id[800],
                        "stuff_id_801": meta.stuff_dataset_id_to_contiguous_id[801],
                        "sem_seg_root": os.path.relpath(os.path.realpath(meta.sem_seg_root), os.path.realpath(DATA_ROOT))},
               "input_opts": [list(x) if isinstance(x, tuple) else x for x in DATA_INPUT_OPTS],
               "training_sampler_seed11_size7": list(itertools.islice(iter(TrainingSampler(7, seed=11)), 30)),
               "inference_shards_10_3": [list(InferenceSampler._get_local_indices(10, 3, r)) for r in range(3)]}
    arrays, cases = {}, []
    train_mapper, test_mapper = DatasetMapper(cfg, True), DatasetMapper(cfg, False)
    for idx in range(len(dicts)):
        for seed in (0, 1, 2, 3, 4):
            np.random.seed(1000 * idx + seed)
            out = train_mapper(dicts[idx])
            key = "train_%d_%d" % (idx, seed)
            inst = out["instances"]
            arrays[key + "_image"] = out["image"].numpy()
            arrays[key + "_sem_seg"] = out["sem_seg"].numpy().astype(np.uint8)
            arrays[key + "_boxes"] = inst.gt_boxes.tensor.numpy()
            arrays[key + "_classes"] = inst.gt_classes.numpy()
            if inst.has("gt_masks"):  # an image whose only annotation is a crowd region yields no mask field
                arrays[key + "_masks"] = np.packbits(inst.gt_masks.tensor.numpy(), axis=-1)
            cases.append({"key": key, "index": idx, "np_seed": 1000 * idx + seed, "image_size": list(inst.image_size),
                          "height": out["height"], "width": out["width"], "image_id": out["image_id"],
                          "keys": sorted(out.keys()), "instance_fields": sorted(inst.get_fields().keys())})
        out = test_mapper(dicts[idx])
        key = "test_%d" % idx
        arrays[key + "_image"] = out["image"].numpy()
        arrays[key + "_sem_seg"] = out["sem_seg"].numpy().astype(np.uint8)
        cases.append({"key": key, "index": idx, "keys": sorted(out.keys()), "height": out["height"], "width": out["width"]})
    listing["cases"] = cases
    json.dump(listing, open(os.path.join(HERE, "data_golden.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "data_golden.npz"), **arrays)
    print("wrote data_golden", len(cases), "cases;", sorted({tuple(c.get("image_size", ())) for c in cases}))


def gen_pseudo_panoptic_fixture():
    """Runs the reference script datasets/prepare_ours/generate_pseudo_panoptic.py (runpy, from a temp directory laid out
    the way its relative paths expect) on three small images: overlapping pseudo instances (one completely covered by
    later ones), a semantic class mostly hidden under instances (> 70 %: skipped), an image without pseudo instances.
    Stand-ins: panopticapi's id2rgb / rgb2id (restated), pycocotools decode (u2seg_amd.data.rle), skimage (unused)."""
    import runpy
    import tempfile

    from u2seg_amd.data import rle
    from u2seg_amd.data.pseudo_panoptic import id2rgb, rgb2id

    sys.meta_path.insert(0, _Finder())
    _Finder.ROOTS = _Finder.ROOTS + ("skimage",)
    import panopticapi.utils as PU
    import pycocotools.mask as PM

    PU.id2rgb, PU.rgb2id, PM.decode = id2rgb, rgb2id, rle.decode
    rs = np.random.RandomState(5)
    work = tempfile.mkdtemp()
    ann_root = os.path.join(work, "datasets", "prepare_ours", "u2seg_annotations")
    for d in ("ins_annotations", "semantic_annotations/stego_coco_train_semantic_seg_resized", "panoptic_annotations"):
        os.makedirs(os.path.join(ann_root, d))
    os.makedirs(os.path.join(work, "datasets", "datasets", "panoptic_anns"))
    os.makedirs(os.path.join(work, "datasets", "datasets", "coco", "annotations"))
    sizes = [(40, 56), (48, 32), (30, 30)]
    images = [{"id": 100 + i, "file_name": "%012d.jpg" % (100 + i), "height": h, "width": w} for i, (h, w) in enumerate(sizes)]
    template = {"images": images, "info": {"description": "small"}, "licenses": [{"id": 1}],
                "annotations": [{"file_name": "%012d.png" % im["id"], "image_id": im["id"], "segments_info": []} for im in images]}
    json.dump(template, open(os.path.join(work, "datasets/datasets/panoptic_anns/panoptic_train2017.json"), "w"))
    json.dump({"images": images}, open(os.path.join(work, "datasets/datasets/coco/annotations/instances_train2017.json"), "w"))
    semantic, pseudo = {}, {"annotations": {}}
    with open(os.path.join(ann_root, "semantic_annotations", "coco_train_img_file_names.txt"), "w") as f:
        for im in images:
            f.write(im["file_name"] + "\n")
    for i, (im, (h, w)) in enumerate(zip(images, sizes)):
        sem = rs.randint(0, 27, (h // 8 + 1, w // 8 + 1)).repeat(8, 0).repeat(8, 1)[:h, :w].astype(np.int64)
        yy, xx = np.mgrid[0:h, 0:w]

        def box_inst(x0, y0, bw, bh, cat):
            m = ((xx >= x0) & (xx < x0 + bw) & (yy >= y0) & (yy < y0 + bh)).astype(np.uint8)
            return {"bbox": [x0, y0, bw, bh], "segmentation": rle.encode(m), "category_id": cat, "iscrowd": 0,
                    "area": int(m.sum())}

        if i == 0:
            sem[:16, :24] = 7  # class 8 after the +1 shift: almost entirely under the first instance -> skipped
            insts = [box_inst(0, 0, 24, 15, 12), box_inst(30, 10, 20, 20, 640), box_inst(32, 12, 8, 8, 3),
                     box_inst(32, 12, 8, 4, 77), box_inst(32, 16, 8, 4, 78)]  # the 8x8 box vanishes under the two 8x4 ones
            pseudo["annotations"][str(im["id"])] = {"segments_info": insts}
        elif i == 1:
            pseudo["annotations"][str(im["id"])] = {"segments_info": [box_inst(4, 6, 20, 30, 800), box_inst(10, 20, 18, 25, 1)]}
        semantic[str(i)] = sem
        np.save(os.path.join(ann_root, "semantic_annotations", "stego_coco_train_semantic_seg_resized", "%d.npy" % i), sem)
    json.dump(pseudo, open(os.path.join(ann_root, "ins_annotations", "cocotrain_800_ins_panoptic.json"), "w"))
    inputs = {"template": template, "pseudo": copy.deepcopy(pseudo), "names": [im["file_name"] for im in images]}
    cwd, argv = os.getcwd(), sys.argv
    os.chdir(work)
    sys.argv = ["generate_pseudo_panoptic.py", "--class_num", "800", "--split", "train"]
    try:
        runpy.run_path(os.path.join(REF, "datasets/prepare_ours/generate_pseudo_panoptic.py"), run_name="__main__")
    finally:
        os.chdir(cwd)
        sys.argv = argv
    from PIL import Image

    out_json = json.load(open(os.path.join(ann_root, "panoptic_annotations", "cocotrain_800.json")))
    arrays = {"semantic_%s" % k: v for k, v in semantic.items()}
    for a in out_json["annotations"]:
        png = np.asarray(Image.open(os.path.join(ann_root, "panoptic_annotations", "cocotrain_800", a["file_name"])))
        arrays["ids_" + a["file_name"]] = rgb2id(png)
    inputs["expected"] = out_json
    json.dump(inputs, open(os.path.join(HERE, "pseudo_panoptic_golden.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "pseudo_panoptic_golden.npz"), **arrays)
    print("wrote pseudo_panoptic_golden:", [(a["file_name"], [s_["id"] for s_ in a["segments_info"]]) for a in out_json["annotations"]])


def gen_eval_fixture():
    """The reference's evaluators on a tiny validation set, from a temp working directory (they write their mapping files
    under ./hungarian_matching): COCOEvaluator.do_hangarain_mapping (instance clusters -> categories) and SemSegEvaluator
    in both modes (votes -> semantic_mapping.json -> remapped confusion matrix -> mIoU / fwIoU / mACC / pACC).
    Stand-ins: the pycocotools json index, maskUtils.iou for boxes (cocoapi's bbIou restated in
    u2seg_amd/evaluation/hungarian.py), no OpenCV (boundary IoU off, as the reference does without cv2)."""
    import tempfile

    from PIL import Image

    from u2seg_amd.evaluation.hungarian import box_iou_xywh

    os.environ.setdefault("CLUSTER_NUM", "800")
    work = tempfile.mkdtemp()
    os.chdir(work)
    install_standins()
    install_data_standins()
    import pycocotools.mask as PM

    PM.iou = lambda dt, gt, iscrowd: (np.stack([box_iou_xywh(d, gt) for d in dt]) if len(gt) else [])
    sys.path.insert(0, REF)
    import types as _types

    sys.modules["detectron2._C"] = _types.ModuleType("detectron2._C")
    from detectron2.data import DatasetCatalog, MetadataCatalog
    from detectron2.data.datasets import register_coco_instances
    from detectron2.data.datasets.coco import load_sem_seg
    from detectron2.evaluation import COCOEvaluator, SemSegEvaluator
    from detectron2.structures import Boxes, Instances

    rs = np.random.RandomState(9)
    img_dir, gt_dir = os.path.join(work, "images"), os.path.join(work, "sem_gt")
    os.makedirs(img_dir)
    os.makedirs(gt_dir)
    sizes = [(40, 48), (36, 36), (48, 40)]
    images, anns, aid = [], [], 1
    gt_maps, cat_ids = {}, [1, 2, 3, 17, 18, 44, 62, 90]
    for i, (h, w) in enumerate(sizes):
        name = "%06d" % (i + 1)
        Image.fromarray(rs.randint(0, 255, (h, w, 3)).astype(np.uint8)).save(os.path.join(img_dir, name + ".jpg"))
        gt = rs.randint(0, 54, (h // 6 + 1, w // 6 + 1)).repeat(6, 0).repeat(6, 1)[:h, :w].astype(np.uint8)
        gt[rs.rand(h, w) < 0.05] = 255
        Image.fromarray(gt, mode="L").save(os.path.join(gt_dir, name + ".png"))
        gt_maps[name] = gt
        images.append({"id": i + 1, "file_name": name + ".jpg", "height": h, "width": w})
        for k in range(4):
            bw, bh = rs.uniform(8, 20), rs.uniform(8, 20)
            x0, y0 = rs.uniform(0, w - bw), rs.uniform(0, h - bh)
            anns.append({"id": aid, "image_id": i + 1, "category_id": cat_ids[(i * 3 + k) % len(cat_ids)], "iscrowd": 0,
                         "bbox": [float(x0), float(y0), float(bw), float(bh)], "area": float(bw * bh)})
            aid += 1
    cats_json = [{"id": c, "name": str(c), "supercategory": "x"} for c in range(1, 91)]
    json_file = os.path.join(work, "val.json")
    json.dump({"images": images, "annotations": anns, "categories": cats_json}, open(json_file, "w"))
    register_coco_instances("tiny_val", {}, json_file, img_dir)
    DatasetCatalog.get("tiny_val")  # fills thing_dataset_id_to_contiguous_id
    DatasetCatalog.register("tiny_val_sem", lambda: load_sem_seg(gt_dir, img_dir))
    MetadataCatalog.get("tiny_val_sem").set(stuff_classes=[str(c) for c in range(28)], ignore_label=255)

    # predictions: per image 7 detections = jittered copies of the ground-truth boxes + strays; cluster ids 0..299
    preds_in, outputs, inputs = [], [], []
    cluster_of = {c: [11 * (j + 1), 11 * (j + 1) + 100] for j, c in enumerate(cat_ids)}
    for im in images:
        boxes, scores, classes = [], [], []
        for a in [a for a in anns if a["image_id"] == im["id"]]:
            x, y, bw, bh = a["bbox"]
            for rep in range(2):
                j = rs.uniform(-1.0, 1.0, 4) * (0.6 if rep == 0 else 3.5)
                boxes.append([x + j[0], y + j[1], x + bw + j[2], y + bh + j[3]])
                scores.append(float(rs.choice([0.95, 0.8, 0.62, 0.55, 0.3])))
                classes.append(int(cluster_of[a["category_id"]][rep] if rs.rand() < 0.8 else rs.randint(0, 300)))
        boxes.append([1.0, 1.0, 6.0, 6.0])
        scores.append(0.9)
        classes.append(299)
        inst = Instances((im["height"], im["width"]))
        inst.pred_boxes = Boxes(torch.tensor(boxes, dtype=torch.float32))
        inst.scores = torch.tensor(scores, dtype=torch.float32)
        inst.pred_classes = torch.tensor(classes, dtype=torch.int64)
        logits = torch.from_numpy(rs.randn(28, im["height"] // 6 + 1, im["width"] // 6 + 1).astype(np.float32))
        logits = logits.repeat_interleave(6, 1).repeat_interleave(6, 2)[:, : im["height"], : im["width"]].contiguous()
        # make the prediction correlate with the ground truth so that votes exist: class (gt mod 27) + 1 gets a bonus
        gt = torch.from_numpy(gt_maps[im["file_name"][:-4]].astype(np.int64))
        bonus = torch.zeros_like(logits)
        bonus.scatter_(0, ((gt % 27) + 1).clamp(max=27)[None], 2.5)
        logits = logits + bonus * (gt != 255)[None]
        outputs.append({"instances": inst, "sem_seg": logits})
        inputs.append({"image_id": im["id"], "file_name": os.path.join(img_dir, im["file_name"]), "height": im["height"],
                       "width": im["width"]})
        preds_in.append({"boxes": boxes, "scores": scores, "classes": classes})
    ev = COCOEvaluator("tiny_val", distributed=False, output_dir=None, mode="hungarian_matching")
    ev.reset()
    ev.process(inputs, outputs)
    import itertools

    coco_results = list(itertools.chain(*[x["instances"] for x in ev._predictions]))
    inst_map = ev.do_hangarain_mapping(300, copy.deepcopy(coco_results), save_path=ev.hungarain_matching_save_path)
    def load_gt(filename, dtype=None):  # the reference's loader passes copy=False, which NumPy 2 rejects when a copy is needed
        return np.array(Image.open(filename), dtype=dtype)

    sem = SemSegEvaluator("tiny_val_sem", distributed=False, output_dir=None, mode="hungarian_matching",
                          sem_seg_loading_fn=load_gt)
    sem._compute_boundary_iou = False
    sem.reset()
    sem.process(inputs, outputs)
    sem.evaluate()
    sem_map = json.load(open("./hungarian_matching/semantic_mapping.json"))
    votes = sorted(zip(sem.pred_det_cate, sem.pseudo_gt_cate))
    sem2 = SemSegEvaluator("tiny_val_sem", distributed=False, output_dir=None, mode="eval", sem_seg_loading_fn=load_gt)
    sem2._compute_boundary_iou = False
    sem2.reset()
    sem2.process(inputs, outputs)
    res = sem2.evaluate()["sem_seg"]
    # panoptic predictions: things carry cluster ids (0..299), stuff the unsupervised classes (1..27); both evaluator modes
    import io

    import panopticapi.utils as PU
    from detectron2.evaluation import COCOPanopticEvaluator
    from u2seg_amd.data.pseudo_panoptic import id2rgb, rgb2id

    PU.id2rgb, PU.rgb2id = id2rgb, rgb2id
    MetadataCatalog.get("tiny_val_pan").set(
        thing_dataset_id_to_contiguous_id=dict(MetadataCatalog.get("tiny_val").thing_dataset_id_to_contiguous_id),
        panoptic_json="unused.json", panoptic_root="unused")
    pan_inputs, pan_outputs, pan_in = [], [], []
    mapped_clusters = [int(k) for k, v in inst_map.items() if v != -1]
    for im in images:
        h, w = im["height"], im["width"]
        ids = np.zeros((h, w), dtype=np.int32)
        segs = []
        blocks = [(0, h // 2, 0, w // 2), (0, h // 2, w // 2, w), (h // 2, h, 0, w // 3), (h // 2, h, w // 3, w)]
        cats = [(True, mapped_clusters[im["id"] % len(mapped_clusters)]), (True, 298), (False, 3 + im["id"]), (False, 27)]
        for sid, ((y0, y1, x0, x1), (thing, cat)) in enumerate(zip(blocks, cats), start=1):
            ids[y0:y1, x0:x1] = sid
            seg = {"id": sid, "isthing": thing, "category_id": cat}
            if thing:
                seg.update(score=0.9, instance_id=sid)
            else:
                seg.update(area=int((y1 - y0) * (x1 - x0)))
            segs.append(seg)
        pan_in.append({"ids": ids.tolist(), "segments_info": copy.deepcopy(segs)})
        pan_inputs.append({"image_id": im["id"], "file_name": os.path.join(img_dir, im["file_name"])})
        pan_outputs.append({"panoptic_seg": (torch.from_numpy(ids), segs)})

    def decoded(ev):
        return [{"image_id": p["image_id"], "file_name": p["file_name"], "segments_info": p["segments_info"],
                 "ids": rgb2id(np.asarray(Image.open(io.BytesIO(p["png_string"])))).tolist()} for p in ev._predictions]

    pan_e = COCOPanopticEvaluator("tiny_val_pan", None)  # the mapping files exist in ./hungarian_matching -> eval mode
    assert pan_e.mode == "eval"
    pan_e.reset()
    pan_e.process(pan_inputs, copy.deepcopy(pan_outputs))
    empty = tempfile.mkdtemp()
    os.chdir(empty)
    pan_h = COCOPanopticEvaluator("tiny_val_pan", None)
    assert pan_h.mode == "hungarian_matching"
    pan_h.reset()
    pan_h.process(pan_inputs, copy.deepcopy(pan_outputs))
    os.chdir(work)
    out = {"images": images, "annotations": anns, "categories": cats_json, "predictions": preds_in,
           "panoptic_inputs": pan_in, "panoptic_eval": decoded(pan_e), "panoptic_matching": decoded(pan_h),
           "coco_results": coco_results, "instance_mapping": {str(k): v for k, v in inst_map.items()},
           "instance_mapping_file": json.load(open("./hungarian_matching/instance_mapping.json")),
           "semantic_mapping_file": sem_map, "semantic_votes": [[int(a), int(b)] for a, b in votes],
           "conf_matrix": sem2._conf_matrix.tolist(),
           "sem_seg_results": {k: (None if v != v else float(v)) for k, v in res.items()},
           "transfer_table": sem.transfer(np.concatenate([np.arange(54), [255]]).astype(int)).tolist()}
    arrays = {"gt_%s" % k: v for k, v in gt_maps.items()}
    for im, o in zip(images, outputs):
        arrays["logits_%s" % im["file_name"][:-4]] = o["sem_seg"].numpy()
    json.dump(out, open(os.path.join(HERE, "eval_golden.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "eval_golden.npz"), **arrays)
    print("wrote eval_golden: instance mapping votes", sum(1 for v in inst_map.values() if v != -1), "semantic",
          sum(1 for v in sem_map.values() if v != -1), "mIoU", res["mIoU"])


def gen_label_prep_fixture():
    """The three remaining scripts of datasets/prepare_ours on small inputs: prepare_stuff_panoptic_fpn.py's function run on
    the pseudo-panoptic golden output, generate_classaware_instanceseg_annotations.py and get_panoptic_anns_supercategory.py
    run as scripts (runpy; the first has absolute paths, so its `open` is redirected to the temp files by base name)."""
    import runpy
    import tempfile

    from PIL import Image

    from u2seg_amd.data.pseudo_panoptic import id2rgb, rgb2id

    sys.meta_path.insert(0, _Finder())
    _Finder.ROOTS = _Finder.ROOTS + ("skimage", "pandas_stub")
    import panopticapi.utils as PU

    PU.id2rgb, PU.rgb2id = id2rgb, rgb2id
    out = {}
    # 1) panoptic -> semantic label maps
    fx = json.load(open(os.path.join(HERE, "pseudo_panoptic_golden.json")))
    arrays = np.load(os.path.join(HERE, "pseudo_panoptic_golden.npz"))
    work = tempfile.mkdtemp()
    pan_root, sem_root = os.path.join(work, "pan"), os.path.join(work, "sem")
    os.makedirs(pan_root)
    json.dump(fx["expected"], open(os.path.join(work, "pan.json"), "w"))
    for a in fx["expected"]["annotations"]:
        Image.fromarray(id2rgb(arrays["ids_" + a["file_name"]])).save(os.path.join(pan_root, a["file_name"]))
    spec = importlib.util.spec_from_file_location("ref_prepare_stuff", os.path.join(REF, "datasets/prepare_ours/prepare_stuff_panoptic_fpn.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_prepare_stuff"] = mod  # its worker pool pickles the per-file function by module name
    spec.loader.exec_module(mod)
    mod.separate_coco_semantic_from_panoptic(os.path.join(work, "pan.json"), pan_root, sem_root, fx["expected"]["categories"])
    sem_arrays = {"sem_" + a["file_name"]: np.asarray(Image.open(os.path.join(sem_root, a["file_name"])))
                  for a in fx["expected"]["annotations"]}
    # 2) class-aware instance annotations
    template = {"licenses": [{"id": 1}], "info": {"year": 2017},
                "images": [{"id": 7, "file_name": "7.jpg"}, {"id": 8, "file_name": "8.jpg"}, {"id": 9, "file_name": "9.jpg"}]}
    masks = [{"ins_id": 0, "image_id": 7, "bbox": [1, 2, 3, 4], "segmentation": {"size": [4, 4], "counts": "04"}, "area": 4},
             {"ins_id": 1, "image_id": 9, "bbox": [0, 0, 2, 2], "segmentation": {"size": [4, 4], "counts": "04"}, "area": 4},
             {"ins_id": 2, "image_id": 7, "bbox": [2, 2, 1, 1], "segmentation": {"size": [4, 4], "counts": "04"}, "area": 1}]
    clusters = {"0.jpg": 17, "1.jpg": 299, "2.jpg": 4}
    files = {"instances_val2017.json": template, "coco_val_usl_dino_800_decode.json": clusters,
             "cutler_cocoval_instances_idx.json": masks}
    d2 = tempfile.mkdtemp()
    for k, v in files.items():
        json.dump(v, open(os.path.join(d2, k), "w"))
    real_open = open

    def redirected(path, *a, **k):
        base = os.path.basename(str(path))
        return real_open(os.path.join(d2, base) if base in files else path, *a, **k)

    cwd = os.getcwd()
    os.chdir(d2)
    try:
        runpy.run_path(os.path.join(REF, "datasets/prepare_ours/generate_classaware_instanceseg_annotations.py"),
                       init_globals={"open": redirected}, run_name="__main__")
        out["classaware"] = {"template": template, "masks": masks, "clusters": clusters,
                             "expected": json.load(open(os.path.join(d2, "uni-training-ann/ins_annotations/cocoval_300.json")))}
        # 3) supercategory version of the ground-truth panoptic json
        d3 = tempfile.mkdtemp()
        os.makedirs(os.path.join(d3, "datasets", "panoptic_anns"))
        os.makedirs(os.path.join(d3, "run"))
        standard = {"images": [{"id": 1}], "categories": [{"id": 1, "name": "person", "isthing": 1},
                                                          {"id": 92, "name": "banner", "isthing": 0},
                                                          {"id": 187, "name": "sky-other-merged", "isthing": 0},
                                                          {"id": 200, "name": "rug-merged", "isthing": 0}],
                    "annotations": [{"image_id": 1, "file_name": "1.png", "segments_info": [
                        {"id": 5, "category_id": 1}, {"id": 6, "category_id": 187}, {"id": 7, "category_id": 92},
                        {"id": 8, "category_id": 200}, {"id": 9, "category_id": 149}]}]}
        json.dump(standard, open(os.path.join(d3, "datasets/panoptic_anns/panoptic_val2017.json"), "w"))
        os.chdir(os.path.join(d3, "run"))
        runpy.run_path(os.path.join(REF, "datasets/prepare_ours/get_panoptic_anns_supercategory.py"), run_name="__main__")
        out["supercategory"] = {"standard": standard, "expected": {
            str(n): json.load(open(os.path.join(d3, "datasets/panoptic_anns/panoptic_val2017_%dsuper.json" % n))) for n in (300, 800)}}
    finally:
        os.chdir(cwd)
    json.dump(out, open(os.path.join(HERE, "label_prep_golden.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "label_prep_golden.npz"), **sem_arrays)
    print("wrote label_prep_golden", {k: np.unique(v).tolist() for k, v in sem_arrays.items()})


def gen_config_fixture():
    """The reference's fully resolved config (its defaults.py + the yaml _BASE_ chain) for the five U2Seg config files."""
    install_standins()
    sys.path.insert(0, REF)
    import types as _types

    sys.modules["detectron2._C"] = _types.ModuleType("detectron2._C")
    from detectron2.config import get_cfg

    def plain(x):
        if isinstance(x, dict):
            return {k: plain(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        return x

    out = {}
    for name in ["u2seg_R50_800", "u2seg_R50_300", "u2seg_eval_800", "u2seg_eval_300"]:
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/%s.yaml" % name))
        out[name] = plain(cfg)
    json.dump(out, open(os.path.join(HERE, "config_golden.json"), "w"))
    print("wrote config_golden", {k: len(v) for k, v in out.items()})


def gen_refunit_fixture():
    """The seeded scenarios of the reference's own unit tests with published expected values - tests/modeling/test_rpn.py:35-67,
    test_fast_rcnn.py:17-45, test_roi_heads.py:39-92 - run through the reference here.  The reference's results are first
    checked against the constants in those tests (so the stand-ins are validated by numbers they did not produce).  The
    seeded module weights are hundreds of MB (fc1 alone is 1024 x 50176), so the fixture keeps the outputs of the heavy
    layers instead - RPN head maps, box predictor outputs, mask logits - together with the reference's sampled proposals;
    the oracle is then held to the same published constants for everything downstream of those layers."""
    import_reference()
    from detectron2.config import get_cfg
    from detectron2.layers import ShapeSpec
    from detectron2.modeling.backbone import build_backbone
    from detectron2.modeling.box_regression import Box2BoxTransform
    from detectron2.modeling.proposal_generator import RPN, build_proposal_generator
    from detectron2.modeling.roi_heads import StandardROIHeads
    from detectron2.modeling.roi_heads.fast_rcnn import FastRCNNOutputLayers
    from detectron2.structures import BitMasks, Boxes, ImageList, Instances
    from detectron2.utils.events import EventStorage

    arrays = {}

    def capture_rpn_head(rpn, prefix):
        def hook(_m, _inp, out):
            arrays[prefix + "objectness"] = out[0][0].detach().numpy().copy()   # single level: [N, A, H, W]
            arrays[prefix + "deltas"] = out[1][0].detach().numpy().copy()       # [N, 4A, H, W]
        rpn.rpn_head.register_forward_hook(hook)

    # ---- test_rpn.py:35-67 ----
    torch.manual_seed(121)
    cfg = get_cfg()
    backbone = build_backbone(cfg)
    rpn = RPN(cfg, backbone.output_shape())
    capture_rpn_head(rpn, "rpn_")
    images = ImageList(torch.rand(2, 20, 30), [(10, 10), (20, 30)])
    features = {"res4": torch.rand(2, 1024, 1, 2)}
    gt = Instances((15, 15))
    gt.gt_boxes = Boxes(torch.tensor([[1, 1, 3, 3], [2, 2, 6, 6]], dtype=torch.float32))
    with EventStorage():
        proposals, losses = rpn(images, features, [gt[0], gt[1]])
    assert torch.allclose(losses["loss_rpn_cls"], torch.tensor(0.08011703193)), losses
    assert torch.allclose(losses["loss_rpn_loc"], torch.tensor(0.101470276)), losses
    assert torch.allclose(proposals[0].proposal_boxes.tensor, torch.tensor([[0, 0, 10, 10], [7.2702, 0, 10, 10]]), atol=1e-4)
    assert torch.allclose(proposals[0].objectness_logits, torch.tensor([0.1596, -0.0007]), atol=1e-4)
    arrays["rpn_proposals1_boxes"] = proposals[1].proposal_boxes.tensor.numpy()
    arrays["rpn_proposals1_logits"] = proposals[1].objectness_logits.numpy()

    # ---- test_fast_rcnn.py:17-45 ----
    torch.manual_seed(132)
    pred = FastRCNNOutputLayers(ShapeSpec(channels=8), box2box_transform=Box2BoxTransform(weights=(10, 10, 5, 5)), num_classes=5)
    pooled = torch.rand(2, 8)
    scores, deltas = pred(pooled)
    prop = Instances((10, 10))
    prop.proposal_boxes = Boxes(torch.tensor([[0.8, 1.1, 3.2, 2.8], [2.3, 2.5, 7, 8]], dtype=torch.float32))
    prop.gt_boxes = Boxes(torch.tensor([[1, 1, 3, 3], [2, 2, 6, 6]], dtype=torch.float32))
    prop.gt_classes = torch.tensor([1, 2])
    with EventStorage():
        fl = pred.losses((scores, deltas), [prop])
    assert torch.allclose(fl["loss_cls"], torch.tensor(1.7951188087)) and torch.allclose(fl["loss_box_reg"], torch.tensor(4.0357131958)), fl
    arrays["fast_scores"], arrays["fast_deltas"] = scores.detach().numpy(), deltas.detach().numpy()

    # ---- test_roi_heads.py:39-92 ----
    torch.manual_seed(121)
    cfg = get_cfg()
    cfg.MODEL.ROI_BOX_HEAD.NAME = "FastRCNNConvFCHead"
    cfg.MODEL.ROI_BOX_HEAD.NUM_FC = 2
    cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE = "ROIAlignV2"
    cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS = (10, 10, 5, 5)
    cfg.MODEL.MASK_ON = True
    images = ImageList(torch.rand(2, 20, 30), [(10, 10), (20, 30)])
    features = {"res4": torch.rand(2, 1024, 1, 2)}
    shape = {"res4": ShapeSpec(channels=1024, stride=16)}
    g0 = Instances((15, 15))
    g0.gt_boxes = Boxes(torch.tensor([[1, 1, 3, 3], [2, 2, 6, 6]], dtype=torch.float32))
    g0.gt_classes = torch.tensor([2, 1])
    g0.gt_masks = BitMasks(torch.rand((2, 15, 15)) > 0.5)
    g1 = Instances((15, 15))
    g1.gt_boxes = Boxes(torch.tensor([[1, 5, 2, 8], [7, 3, 10, 5]], dtype=torch.float32))
    g1.gt_classes = torch.tensor([1, 2])
    g1.gt_masks = BitMasks(torch.rand((2, 15, 15)) > 0.5)
    pg = build_proposal_generator(cfg, shape)
    heads = StandardROIHeads(cfg, shape)
    capture_rpn_head(pg, "heads_rpn_")
    box_losses = heads.box_predictor.losses

    def capture_box(predictions, props):
        arrays["heads_scores"], arrays["heads_deltas"] = predictions[0].detach().numpy().copy(), predictions[1].detach().numpy().copy()
        for i, p in enumerate(props):
            arrays["heads_sampled%d_boxes" % i] = p.proposal_boxes.tensor.numpy().copy()
            arrays["heads_sampled%d_classes" % i] = p.gt_classes.numpy().copy()
            arrays["heads_sampled%d_gt_boxes" % i] = p.gt_boxes.tensor.numpy().copy()
        return box_losses(predictions, props)

    heads.box_predictor.losses = capture_box
    mask_layers = heads.mask_head.layers

    def capture_mask(x):
        out = mask_layers(x)
        arrays["heads_mask_logits"] = out.detach().numpy().copy()
        return out

    heads.mask_head.layers = capture_mask

    def mask_pre(_m, args):
        for i, p in enumerate(args[1]):
            arrays["heads_fg%d_boxes" % i] = p.proposal_boxes.tensor.numpy().copy()
            arrays["heads_fg%d_classes" % i] = p.gt_classes.numpy().copy()
            arrays["heads_fg%d_masks" % i] = p.gt_masks.tensor.numpy().copy()

    heads.mask_head.register_forward_pre_hook(mask_pre)
    with EventStorage():
        proposals, pl = pg(images, features, [g0, g1])
        _, dl = heads(images, features, proposals, [g0, g1])
    dl.update(pl)
    expected = {"loss_cls": 4.5253729820251465, "loss_box_reg": 0.009785720147192478, "loss_mask": 0.693184494972229,
                "loss_rpn_cls": 0.08186662942171097, "loss_rpn_loc": 0.1104838103055954}
    for k, v in expected.items():
        assert torch.allclose(dl[k], torch.tensor(v)), (k, float(dl[k]), v)
    for i, p in enumerate(proposals):
        arrays["heads_proposals%d_boxes" % i] = p.proposal_boxes.tensor.numpy().copy()
        arrays["heads_proposals%d_logits" % i] = p.objectness_logits.numpy().copy()
    arrays["heads_masks0"], arrays["heads_masks1"] = g0.gt_masks.tensor.numpy(), g1.gt_masks.tensor.numpy()
    np.savez_compressed(os.path.join(HERE, "refunit_golden.npz"), **arrays)
    print("wrote refunit_golden: the reference reproduces the constants of its tests;", len(arrays), "arrays,",
          sum(v.size * v.itemsize for v in arrays.values()) // 1024, "KiB raw", {k: v.shape for k, v in arrays.items() if "mask_logits" in k or "sampled0" in k})


def gen_cocoeval_fixture():
    """The reference's C++ COCO evaluation core (layers/csrc/cocoeval/cocoeval.cpp, compiled into the _C extension built by
    import_reference) on a synthetic detection problem: 12 images, 5 categories, crowd regions, boxes of all three area
    classes, 130 detections in one (image, category) cell (budget truncation), tied scores.  Inputs are prepared the way
    fast_eval_api.py:55-88 does (instances + IoU tables per image and category); outputs: the precision / recall / score
    tables of COCOevalEvaluateImages + COCOevalAccumulate."""
    import_reference()
    C = sys.modules["detectron2._C"]
    from u2seg_amd.evaluation import cocoeval as CE

    rs = np.random.RandomState(21)
    images = [{"id": 3 * i + 1, "height": 200, "width": 300} for i in range(12)]
    cats = [{"id": c} for c in (1, 2, 5, 7, 9)]
    anns, results, aid = [], [], 1
    for im in images:
        for _ in range(rs.randint(0, 7)):
            side = rs.choice([12.0, 50.0, 140.0]) * rs.uniform(0.7, 1.3)
            w, h = side, side * rs.uniform(0.6, 1.4)
            x, y = rs.uniform(0, 300 - w), rs.uniform(0, 200 - h)
            cat = int(rs.choice([1, 2, 5, 7, 9]))
            anns.append({"id": aid, "image_id": im["id"], "category_id": cat, "bbox": [float(x), float(y), float(w), float(h)],
                         "area": float(w * h * rs.uniform(0.5, 0.9)), "iscrowd": int(rs.rand() < 0.15)})
            aid += 1
            for _ in range(rs.randint(0, 4)):  # detections around the instance, some in the wrong class
                j = rs.uniform(-1, 1, 4) * side * rs.choice([0.03, 0.15, 0.5])
                results.append({"image_id": im["id"], "category_id": cat if rs.rand() < 0.8 else int(rs.choice([1, 2, 5, 7, 9])),
                                "bbox": [float(x + j[0]), float(y + j[1]), float(max(w + j[2], 1)), float(max(h + j[3], 1))],
                                "score": float(np.round(rs.uniform(0.05, 1.0), 2))})  # two decimals: ties happen
        for _ in range(rs.randint(0, 5)):  # strays
            results.append({"image_id": im["id"], "category_id": int(rs.choice([1, 2, 5, 7, 9])),
                            "bbox": [float(rs.uniform(0, 250)), float(rs.uniform(0, 150)), float(rs.uniform(5, 50)), float(rs.uniform(5, 50))],
                            "score": float(np.round(rs.uniform(0.05, 0.6), 2))})
    for k in range(130):  # one crowded cell beyond the 100-detection budget
        results.append({"image_id": 1, "category_id": 2, "bbox": [float(2 * k % 250), float(k % 150), 30.0, 30.0],
                        "score": float(np.round(rs.uniform(0.01, 0.99), 3))})
    dataset = {"images": images, "annotations": anns, "categories": cats}
    params = CE.Params(sorted(im["id"] for im in images), sorted(c["id"] for c in cats))
    gts, dts = CE.prepare(anns, results, params)
    ious = CE.compute_ious(gts, dts, params)

    def cpp(instances):
        return [C.InstanceAnnotation(int(o["id"]), float(o.get("score", 0.0)), float(o["area"]), bool(o["iscrowd"]),
                                     bool(o["ignore"])) for o in instances]

    gt_cpp = [[cpp(gts.get((i, c), [])) for c in params.catIds] for i in params.imgIds]
    dt_cpp = [[cpp(dts.get((i, c), [])) for c in params.catIds] for i in params.imgIds]
    iou_cpp = [[(ious[i, c].tolist() if len(ious[i, c]) else []) for c in params.catIds] for i in params.imgIds]
    evals = C.COCOevalEvaluateImages(params.areaRng, params.maxDets[-1], params.iouThrs.tolist(), iou_cpp, gt_cpp, dt_cpp)

    class P:
        pass

    pobj = P()
    pobj.recThrs, pobj.maxDets, pobj.iouThrs = params.recThrs.tolist(), params.maxDets, params.iouThrs.tolist()
    pobj.catIds, pobj.areaRng, pobj.imgIds, pobj.useCats = params.catIds, params.areaRng, params.imgIds, 1
    acc = C.COCOevalAccumulate(pobj, evals)
    counts = list(acc["counts"])
    precision = np.array(acc["precision"]).reshape(counts)
    scores = np.array(acc["scores"]).reshape(counts)
    recall = np.array(acc["recall"]).reshape(counts[:1] + counts[2:])
    json.dump({"dataset": dataset, "results": results}, open(os.path.join(HERE, "cocoeval_golden.json"), "w"))
    np.savez_compressed(os.path.join(HERE, "cocoeval_golden.npz"), precision=precision, recall=recall, scores=scores)
    print("wrote cocoeval_golden:", len(anns), "gt,", len(results), "detections; mean precision over valid entries",
          float(precision[precision > -1].mean()))


def _import_nn_utils():
    for m in ["pykeops", "pykeops.torch", "torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.models",
              "yacs", "yacs.config", "termcolor", "clip"]:
        sys.modules.setdefault(m, mock.MagicMock())
    pkg = types.ModuleType("u")
    pkg.__path__ = []
    sys.modules["u"] = pkg
    cu = types.ModuleType("u.config_utils")
    cu.cfg, cu.logger = mock.MagicMock(), mock.MagicMock()
    sys.modules["u.config_utils"] = cu
    path = os.path.join(REF, "u2seg/Instance_Clustering/shared/utils/nn_utils.py")
    src = open(path).read().replace("from .config_utils", "from u.config_utils")
    mod = types.ModuleType("u.nn_utils")
    mod.__package__ = "u"
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


class _DenseLazyTensor:
    """Stand-in for pykeops.torch.LazyTensor (pykeops is not in this image), covering exactly the expression kNN builds
    (nn_utils.py:210-216): broadcasting `-`, `** 2`, `.sum(-1)` evaluated densely in fp32, and Kmin_argKmin(K, dim=1) =
    the K smallest values of every row in ascending order with their column indices (pykeops' documented reduction;
    equal values keep the smaller column first here, pykeops leaves that order unspecified)."""

    def __init__(self, t):
        self.t = t

    def __sub__(self, other):
        return _DenseLazyTensor(self.t - other.t)

    def __pow__(self, e):
        return _DenseLazyTensor(self.t ** e)

    def sum(self, dim):
        return _DenseLazyTensor(self.t.sum(dim))

    def Kmin_argKmin(self, K, dim, backend=None):
        assert dim == 1
        v, i = torch.sort(self.t, dim=1, stable=True)
        return v[:, :K].contiguous(), i[:, :K].contiguous()


def gen_knn_fixture():
    """Reference partitioned_kNN (nn_utils.py:230-299): its own partition loop and argsort merge over 3 partitions
    (300 + 300 + 100 rows), with pykeops' LazyTensor served by _DenseLazyTensor and .cuda() made the identity."""
    mod = _import_nn_utils()
    mod.LazyTensor = _DenseLazyTensor
    mod.save_npy = lambda *a, **k: None
    g = torch.Generator().manual_seed(11)
    N, D, K = 700, 32, 20
    centers = torch.randn((9, D), generator=g)
    x = centers[torch.randint(0, 9, (N,), generator=g)] + 0.3 * torch.randn((N, D), generator=g)
    x = torch.nn.functional.normalize(x, dim=1)
    x[650] = x[5]      # exact duplicates across partitions: tied distances, both at 0 from each other
    x[310] = x[305]
    with mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self):
        d_knns, ind_knns = mod.partitioned_kNN(x, K=K, recompute=True, partitions_size=300)
        ind_one, d_one = mod.kNN(x, x, K=K)
    assert torch.equal(d_knns, d_one)
    # the representative of every cluster (nn_utils.py:408-439) from the density d_knns.mean(1), on labels with an empty
    # cluster and tied densities (the planted twin rows)
    gl = torch.Generator().manual_seed(12)
    labels = torch.randint(0, 14, (N,), generator=gl)
    labels[labels == 6] = 5  # cluster 6 stays empty
    neighbors_dist = d_knns.mean(dim=1)
    with mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self):
        sel_all = mod.get_selection_without_reg(labels, neighbors_dist, 14, final_sample_num=13)
        sel_cut = mod.get_selection_without_reg(labels, neighbors_dist, [9, 2, 6, 0, 13], final_sample_num=3)
    np.savez_compressed(os.path.join(HERE, "knn_golden.npz"), x=x.numpy(), d_knns=d_knns.numpy(), ind_knns=ind_knns.numpy(),
                        K=np.array(K), partitions_size=np.array(300), sel_labels=labels.numpy(), sel_all=np.asarray(sel_all),
                        sel_cut=np.asarray(sel_cut))
    print("wrote knn_golden.npz", d_knns.shape, float(d_knns.mean()))


def gen_kmeans_fixture():
    """Reference KMeans (u2seg/Instance_Clustering/shared/utils/nn_utils.py:304-379) through its plain-torch branch."""
    mod = _import_nn_utils()
    g = torch.Generator().manual_seed(3)
    K, D, N = 12, 64, 3000
    centers = torch.randn((K, D), generator=g) * 3
    x = centers[torch.randint(0, K, (N,), generator=g)] + 0.5 * torch.randn((N, D), generator=g)
    seed = 0
    torch.manual_seed(seed)
    init = torch.randperm(N)[:K]  # what KMeans draws first after manual_seed(seed) (:337-340)
    cl, c = mod.KMeans(x, seed=seed, K=K, Niter=6, verbose=False, force_no_lazy_tensor=True)
    np.savez_compressed(os.path.join(HERE, "kmeans_golden.npz"), x=x.numpy(), init=init.numpy(), labels=cl.numpy(),
                        centroids=c.numpy(), niter=np.array(6))
    print("wrote kmeans_golden.npz", cl.shape, c.shape)


def gen_bf16_units_fixture():
    """Pins the oracle's bf16 mode (OracleModel(emulate_bf16=True)): the REFERENCE's PanopticFPN, name-keyed weights, one
    synthetic 96 x 128 image, train mode, run under torch.autocast("cpu", dtype=torch.bfloat16) - the precision recipe of
    engine/train_loop.py:451-521 with the dtype BASELINE.json asks for.  Forward hooks keep, for a handful of units that
    together contain every layer type of the conv stack, the unit's bf16 INPUT and OUTPUT (teacher forcing: the oracle gets the
    reference's own input, so only this unit's rounding points are compared, not 50 layers of amplified noise), plus the ten
    losses of the run.  tests/test_oracle_golden.py::test_bf16_mode_vs_reference_autocast holds the oracle to them."""
    import_reference()
    os.environ.setdefault("CLUSTER_NUM", "800")
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.utils.events import EventStorage

    from u2seg_amd.data import make_synthetic_batch

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.WEIGHTS = ""
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v))
    model.train()
    bu = model.backbone.bottom_up
    units = {"stem": bu.stem, "res2.0": bu.res2[0], "res3.0": bu.res3[0], "res4.1": bu.res4[1], "res5.2": bu.res5[2],
             "fpn_lateral4": model.backbone.fpn_lateral4, "fpn_output3": model.backbone.fpn_output3,
             "sem.p3.0": model.sem_seg_head.scale_heads[1][0], "sem.predictor": model.sem_seg_head.predictor,
             "rpn.conv": model.proposal_generator.rpn_head.conv,
             "rpn.objectness": model.proposal_generator.rpn_head.objectness_logits,
             "box.fc2": model.roi_heads.box_head[0].fc2, "mask.fcn1": model.roi_heads.mask_head.mask_fcn1}
    got = {}

    def hook(name):
        def fn(_m, inp, out):
            if name not in got:  # modules shared across FPN levels (rpn head) fire once per level: keep the first (p2)
                i, o = inp[0].detach(), out.detach()
                if name in ("box.fc2", "mask.fcn1"):  # per-ROI independent units (no normalisation): a few ROIs are enough
                    keep = 64 if name == "box.fc2" else 4
                    i, o = i[:keep], o[:keep]
                got[name] = (i.clone(), o.clone())
        return fn

    for k, m in units.items():
        m.register_forward_hook(hook(k))
    batch = to_ref_batch(make_synthetic_batch(1, height=96, width=128))
    torch.manual_seed(5)
    with EventStorage(), torch.autocast("cpu", dtype=torch.bfloat16):
        losses = model(batch)
    arrays = {}
    meta = {"image_hw": [96, 128], "num_images": 1, "seed": 5, "autocast": "cpu bfloat16",
            "losses": {k: float(v) for k, v in losses.items()}, "dtypes": {}}
    for k, (i, o) in got.items():
        meta["dtypes"][k] = [str(i.dtype), str(o.dtype)]
        for tag, t in (("in", i), ("out", o)):
            if t.dtype == torch.bfloat16:
                arrays["%s.%s" % (k, tag)] = t.contiguous().view(torch.int16).numpy().copy()
            else:
                arrays["%s.%s" % (k, tag)] = t.float().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "bf16_units_golden.npz"), **arrays)
    json.dump(meta, open(os.path.join(HERE, "bf16_units_golden.json"), "w"), indent=1)
    print("wrote bf16_units_golden", {k: list(v.shape) for k, v in arrays.items()}, meta["losses"], meta["dtypes"])


def gen_checkpoint_matching_fixture():
    """detectron2/checkpoint/c2_model_loading.py:209-330 (align_and_update_state_dicts, the name-matching heuristic behind
    DetectionCheckpointer._load_model for files with `matching_heuristics`, i.e. U2Seg's dino_RN50_pretrain_d2_format.pkl): the
    REFERENCE function run on the reference model's own 431 state-dict keys against a backbone-only, prefix-free d2-format key
    set (+ an unused classifier weight, + one tensor of the wrong shape).  Tensors are 1-element stand-ins tagged with their
    index except where a shape matters.  fvcore's Checkpointer base class is absent here, so the file round trip itself
    (pickle / torch.save) cannot be produced by the reference; what is pinned is its matching logic on the real key set."""
    import_reference()
    os.environ.setdefault("CLUSTER_NUM", "800")
    from detectron2.checkpoint.c2_model_loading import align_and_update_state_dicts
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.WEIGHTS = ""
    model_sd = build_model(cfg).state_dict()
    prefix = "backbone.bottom_up."
    ckpt = {}
    for i, (k, v) in enumerate(model_sd.items()):
        if k.startswith(prefix) and "num_batches_tracked" not in k:
            ckpt[k[len(prefix):]] = torch.full(tuple(v.shape), float(i))
    ckpt["stem.fc.weight"] = torch.zeros(1000, 2048)            # present in ImageNet-style files, unused by the detector
    ckpt["res2.0.conv1.weight"] = torch.zeros(3, 3)              # wrong shape: skipped with a warning
    shapes = {k: list(v.shape) for k, v in ckpt.items()}
    out = align_and_update_state_dicts(model_sd, dict(ckpt), c2_conversion=False)
    tag = {id(v): k for k, v in ckpt.items()}
    mapping = {mk: tag[id(v)] for mk, v in out.items() if id(v) in tag}
    json.dump({"model_keys": {k: list(v.shape) for k, v in model_sd.items()}, "ckpt_shapes": shapes, "result": mapping},
              open(os.path.join(HERE, "checkpoint_matching_golden.json"), "w"), indent=0)
    print("wrote checkpoint_matching_golden.json:", len(mapping), "entries,", sum(1 for a, b in mapping.items() if a != b), "renamed")


REDUCED_MODEL_OPTS = ["MODEL.RESNETS.STEM_OUT_CHANNELS", 8, "MODEL.RESNETS.RES2_OUT_CHANNELS", 16, "MODEL.RESNETS.WIDTH_PER_GROUP", 4,
                      "MODEL.FPN.OUT_CHANNELS", 32, "MODEL.ROI_BOX_HEAD.FC_DIM", 48, "MODEL.ROI_MASK_HEAD.CONV_DIM", 32,
                      "MODEL.SEM_SEG_HEAD.CONVS_DIM", 32, "MODEL.ROI_HEADS.NUM_CLASSES", 6, "MODEL.SEM_SEG_HEAD.NUM_CLASSES", 5]


def gen_checkpoint_files_fixture():
    """files in the reference's two ON-DISK forms, written from the reference's own model object.
      * checkpoint_small.pth - what fvcore's Checkpointer.save writes (checkpoint/detection_checkpoint.py:17-143 on top of it):
        torch.save({"model": model.state_dict(), "optimizer": ..., "scheduler": ..., "iteration": ...}); model, optimizer and
        LR scheduler are the reference's (build_model / build_optimizer / WarmupMultiStepLR), the model is u2seg_R50_800 with
        the widths reduced through its own config keys so that the file stays small (every layer type and every one of the
        431 state-dict keys is present);
      * checkpoint_small_d2.pkl - the Detectron2 model-zoo pickle form of U2Seg's dino_RN50_pretrain_d2_format.pkl
        (u2seg_R50_800.yaml:6): {"model": {backbone names without prefix: ndarray}, "__author__", "matching_heuristics": True};
      * checkpoint_files_golden.json - crc32 of every tensor of the .pth by model key, and for the .pkl the key -> source key
        assignment the reference's own align_and_update_state_dicts makes on this model."""
    import pickle

    import_reference()
    os.environ.setdefault("CLUSTER_NUM", "800")
    from detectron2.checkpoint.c2_model_loading import align_and_update_state_dicts
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.solver import build_optimizer
    from detectron2.solver.lr_scheduler import WarmupMultiStepLR

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.WEIGHTS", ""] + REDUCED_MODEL_OPTS)
    torch.manual_seed(11)
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if v.dtype.is_floating_point:
                v.copy_(det_fill(k, v))
    opt = build_optimizer(cfg, model)
    # (the fvcore composite build_lr_scheduler assembles is absent here; the reference's own plain-python scheduler class)
    sched = WarmupMultiStepLR(opt, list(cfg.SOLVER.STEPS), cfg.SOLVER.GAMMA, cfg.SOLVER.WARMUP_FACTOR, cfg.SOLVER.WARMUP_ITERS,
                              cfg.SOLVER.WARMUP_METHOD)
    sd = model.state_dict()
    torch.save({"model": sd, "optimizer": opt.state_dict(), "scheduler": sched.state_dict(), "iteration": 1234},
               os.path.join(HERE, "checkpoint_small.pth"))
    prefix = "backbone.bottom_up."
    zoo = {k[len(prefix):]: v.numpy().copy() + 1.0 for k, v in sd.items() if k.startswith(prefix) and "num_batches_tracked" not in k}
    zoo["stem.fc.weight"] = np.zeros((10, 2048), dtype=np.float32)   # ImageNet classifier left-over: reported, not loaded
    with open(os.path.join(HERE, "checkpoint_small_d2.pkl"), "wb") as f:
        pickle.dump({"model": zoo, "__author__": "make_fixtures.py (reference model, reduced widths)", "matching_heuristics": True}, f)
    as_t = {k: torch.from_numpy(v) for k, v in zoo.items()}
    out = align_and_update_state_dicts(dict(sd), dict(as_t), c2_conversion=False)
    tag = {id(v): k for k, v in as_t.items()}
    mapping = {mk: tag[id(v)] for mk, v in out.items() if id(v) in tag}
    crc = {k: zlib.crc32(v.contiguous().numpy().tobytes()) for k, v in sd.items()}
    json.dump({"opts": REDUCED_MODEL_OPTS, "pth_crc32": crc, "pth_shapes": {k: list(v.shape) for k, v in sd.items()},
               "pkl_assignment": mapping, "iteration": 1234},
              open(os.path.join(HERE, "checkpoint_files_golden.json"), "w"), indent=0)
    print("wrote checkpoint_small.pth (%d keys, %.2f MB), checkpoint_small_d2.pkl (%d arrays)" % (
        len(sd), os.path.getsize(os.path.join(HERE, "checkpoint_small.pth")) / 1e6, len(zoo)))


def gen_resume_fixture(hw=(192, 256), nimg=2, seed=9, steps=4, save_after=1):
    """SURVEY 8(f) row 2, the --resume leg: a checkpoint written by the REFERENCE in the middle of a run, and how that run went
    on.  The reference's PanopticFPN (u2seg_R50_800, widths reduced through its own config keys: the file stays at a few MB),
    its build_optimizer (per-parameter clip around torch.optim.SGD) and its WarmupMultiStepLR take `steps` steps with the
    compressed schedule of the trajectory fixture; after step `save_after` the state is saved in the nesting DefaultTrainer's
    checkpointer writes (engine/defaults.py:389-394,499-506, engine/train_loop.py:195-208,423-430, engine/hooks.py:365-367:
    the trainer itself is the checkpointable, PeriodicCheckpointer adds iteration=...):
        {"model", "trainer": {"iteration", "hooks": {"LRScheduler": scheduler.state_dict()},
                              "_trainer": {"iteration", "optimizer": optimizer.state_dict()}}, "iteration"}
    - every value is a state_dict() of a reference object; the momentum buffers are non-trivial (two steps of history).
      * checkpoint_resume.pth   - that file;
      * resume_golden.json      - torch's parameter numbering -> parameter name (the grouping of reduce_param_groups), crc32 of every
                                  momentum buffer and saved weight by name, lr / losses of every step, where the parameters ended
                                  up after the last step."""
    import_reference()
    os.environ.setdefault("CLUSTER_NUM", "800")
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.solver import build_optimizer
    from detectron2.solver.lr_scheduler import WarmupMultiStepLR
    from detectron2.utils.events import EventStorage

    from u2seg_amd.data import make_synthetic_batch

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.WEIGHTS", ""] + REDUCED_MODEL_OPTS + list(TRAJ_OVERRIDES))
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v))
    model.train()
    opt = build_optimizer(cfg, model)
    sched = WarmupMultiStepLR(opt, list(cfg.SOLVER.STEPS), cfg.SOLVER.GAMMA, cfg.SOLVER.WARMUP_FACTOR,
                              cfg.SOLVER.WARMUP_ITERS, cfg.SOLVER.WARMUP_METHOD)
    names = {id(p): k for k, p in model.named_parameters()}
    numbering, n = {}, 0
    for g in opt.param_groups:
        for p in g["params"]:
            numbering[n] = names[id(p)]
            n += 1
    picked = ["backbone.bottom_up.stem.conv1.weight", "backbone.bottom_up.res4.5.conv3.norm.weight", "backbone.fpn_output2.weight",
              "roi_heads.box_predictor.1.cls_score.weight", "roi_heads.mask_head.deconv.weight", "sem_seg_head.predictor.bias"]
    torch.manual_seed(seed)
    per_step, lrs, saved, at_save = [], [], None, None
    with EventStorage() as storage:
        for it in range(steps):
            batch = to_ref_batch(make_synthetic_batch(nimg, height=hw[0], width=hw[1], start_index=it * nimg,
                                                      num_thing_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                                                      num_stuff_classes=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES))
            losses = model(batch)
            opt.zero_grad()
            sum(losses.values()).backward()
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
            storage.step()
            per_step.append({k: float(v) for k, v in losses.items()})
            print("step", it, "lr", lrs[-1], "total", sum(per_step[-1].values()))
            if it == save_after:
                osd, ssd = opt.state_dict(), sched.state_dict()
                saved = {"model": {k: v.clone() for k, v in model.state_dict().items()},
                         "trainer": {"iteration": it, "hooks": {"LRScheduler": ssd},
                                     "_trainer": {"iteration": it, "optimizer": {
                                         "state": {i: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                                                   for i, st in osd["state"].items()},
                                         "param_groups": osd["param_groups"]}}},
                         "iteration": it}
                torch.save(saved, os.path.join(HERE, "checkpoint_resume.pth"))
                at_save = {k: dict(model.named_parameters())[k].detach().clone() for k in picked}
                # the sampler draws of the following steps come from torch's global CPU stream: the state it has HERE
                rng_after_save = torch.get_rng_state()
    params = dict(model.named_parameters())
    crc = lambda t: zlib.crc32(t.detach().contiguous().numpy().tobytes())
    mom = {numbering[i]: crc(st["momentum_buffer"]) for i, st in saved["trainer"]["_trainer"]["optimizer"]["state"].items()}
    out = {"opts": REDUCED_MODEL_OPTS + [list(x) if isinstance(x, tuple) else x for x in TRAJ_OVERRIDES], "image_hw": list(hw),
           "num_images": nimg, "seed": seed, "steps": steps, "saved_iteration": save_after,
           "num_thing_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES, "num_stuff_classes": cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES, "lr": lrs, "losses": per_step,
           "numbering": {str(i): k for i, k in numbering.items()},
           "group_sizes": [len(g["params"]) for g in opt.param_groups],
           "group_weight_decay": [g["weight_decay"] for g in opt.param_groups],
           "momentum_crc32": mom, "model_crc32": {k: crc(v) for k, v in saved["model"].items()},
           "momentum_norm": {numbering[i]: float(st["momentum_buffer"].double().norm())
                             for i, st in saved["trainer"]["_trainer"]["optimizer"]["state"].items() if numbering[i] in picked},
           "rng_state_after_save_b64": __import__("base64").b64encode(rng_after_save.numpy().tobytes()).decode(),
           "param_norm": {k: float(params[k].double().norm()) for k in picked},
           "param_delta_norm_since_save": {k: float((params[k].detach() - at_save[k]).double().norm()) for k in picked},
           "num_batches_tracked": int(dict(model.named_buffers())["backbone.bottom_up.stem.conv1.norm.num_batches_tracked"])}
    json.dump(out, open(os.path.join(HERE, "resume_golden.json"), "w"), indent=0)
    print("wrote checkpoint_resume.pth (%.2f MB) and resume_golden.json" % (os.path.getsize(os.path.join(HERE, "checkpoint_resume.pth")) / 1e6))


def fullwidth_grad(name, tensor):
    """Synthetic gradient of the full-width checkpoint fixture (shared with the test): small enough that the per-parameter clip
    (CLIP_VALUE 1.0) never scales it, so the step is plain SGD arithmetic."""
    return det_fill(name + "#grad", tensor) * 1e-3


def gen_fullwidth_fixture():
    """the FULL-WIDTH u2seg_R50_800 (RES2_OUT_CHANNELS 256, 800 + 1 classes, 76 M parameters) as the
    reference checkpoints it.  A 600 MB file cannot be a fixture, its content can: every tensor is det_fill(name), the reference's
    build_optimizer (clip-wrapped torch.optim.SGD) takes ONE step on the gradients fullwidth_grad(name) and its WarmupMultiStepLR one
    step; recorded are the crc32 of every model tensor and momentum buffer after that step, the parameter groups and torch's
    numbering.  The test rebuilds the same file content with torch alone (same fills, a plain torch.optim.SGD over the recorded
    grouping), proves it equal by these crc32, saves it in the reference's nesting and loads it on the device."""
    import_reference()
    os.environ["CLUSTER_NUM"] = "800"
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.solver import build_optimizer
    from detectron2.solver.lr_scheduler import WarmupMultiStepLR

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.WEIGHTS", ""])
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v))
    opt = build_optimizer(cfg, model)
    sched = WarmupMultiStepLR(opt, list(cfg.SOLVER.STEPS), cfg.SOLVER.GAMMA, cfg.SOLVER.WARMUP_FACTOR,
                              cfg.SOLVER.WARMUP_ITERS, cfg.SOLVER.WARMUP_METHOD)
    names = {id(p): k for k, p in model.named_parameters()}
    lr0 = opt.param_groups[0]["lr"]
    for k, p in model.named_parameters():
        p.grad = fullwidth_grad(k, p)
        assert float(p.grad.norm()) < 0.9 * cfg.SOLVER.CLIP_GRADIENTS.CLIP_VALUE, (k, float(p.grad.norm()))
    opt.step()
    sched.step()
    crc = lambda t: zlib.crc32(t.detach().contiguous().numpy().tobytes())
    osd = opt.state_dict()
    numbering = [names[id(p)] for g in opt.param_groups for p in g["params"]]
    out = {"num_parameters": sum(p.numel() for p in model.parameters()), "lr_of_the_step": lr0,
           "param_groups": [{k: v for k, v in g.items() if k != "params"} | {"n": len(g["params"])} for g in osd["param_groups"]],
           "numbering": numbering, "scheduler": {k: v for k, v in sched.state_dict().items() if isinstance(v, (int, float, list))},
           "model_crc32": {k: crc(v) for k, v in model.state_dict().items()},
           "momentum_crc32": {numbering[i]: crc(st["momentum_buffer"]) for i, st in osd["state"].items()},
           "torch": torch.__version__}
    json.dump(out, open(os.path.join(HERE, "fullwidth_checkpoint_golden.json"), "w"), indent=0)
    print("wrote fullwidth_checkpoint_golden.json: %d tensors, %d parameters, groups %s" %
          (len(out["model_crc32"]), out["num_parameters"], [g["n"] for g in out["param_groups"]]))


def gen_param_groups_fixture():
    """the reference groups parameters by their per-parameter override DICT, not by the weight-decay value
    (solver/build.py:123-129,181-236,255-279).  For several (WEIGHT_DECAY, WEIGHT_DECAY_NORM, WEIGHT_DECAY_BIAS) triples -
    including overrides that EQUAL the default - record what the reference's build_optimizer forms on the reduced
    u2seg_R50_800: group sizes, each group's weight decay, and a crc32 of torch's parameter numbering (names in group order)."""
    import_reference()
    os.environ.setdefault("CLUSTER_NUM", "800")
    from detectron2.config import get_cfg
    from detectron2.modeling import build_model
    from detectron2.solver import build_optimizer

    cases = []
    for wd, wdn, wdb in [(1e-4, 0.0, None), (1e-4, 1e-4, None), (1e-4, 0.0, 1e-4), (1e-4, 1e-4, 1e-4), (1e-4, 0.0, 0.0),
                         (5e-5, 1e-4, 5e-5), (1e-4, None, None), (1e-4, None, 0.0)]:
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(REF, "configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml"))
        cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.WEIGHTS", ""] + REDUCED_MODEL_OPTS)
        cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_NORM, cfg.SOLVER.WEIGHT_DECAY_BIAS = wd, wdn, wdb
        model = build_model(cfg)
        opt = build_optimizer(cfg, model)
        names = {id(p): k for k, p in model.named_parameters()}
        order = [names[id(p)] for g in opt.param_groups for p in g["params"]]
        cases.append({"weight_decay": wd, "weight_decay_norm": wdn, "weight_decay_bias": wdb,
                      "group_sizes": [len(g["params"]) for g in opt.param_groups],
                      "group_weight_decay": [g["weight_decay"] for g in opt.param_groups],
                      "first_names": [names[id(g["params"][0])] for g in opt.param_groups],
                      "numbering_crc32": zlib.crc32("\n".join(order).encode())})
        print(cases[-1]["group_sizes"], cases[-1]["group_weight_decay"])
    json.dump({"opts": REDUCED_MODEL_OPTS, "cases": cases}, open(os.path.join(HERE, "param_groups_golden.json"), "w"), indent=0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if a.only in ("", "kmeans"):
        gen_kmeans_fixture()
    if a.only in ("", "knn"):
        gen_knn_fixture()
    if a.only == "cocoeval":
        gen_cocoeval_fixture()
        sys.exit(0)
    if a.only == "refunit":
        gen_refunit_fixture()
        sys.exit(0)
    if a.only == "checkpoint":
        gen_checkpoint_matching_fixture()
        sys.exit(0)
    if a.only == "checkpoint_files":
        gen_checkpoint_files_fixture()
        sys.exit(0)
    if a.only == "resume":
        gen_resume_fixture()
        sys.exit(0)
    if a.only == "fullwidth":
        gen_fullwidth_fixture()
        sys.exit(0)
    if a.only == "param_groups":
        gen_param_groups_fixture()
        sys.exit(0)
    if a.only == "bf16_units":
        gen_bf16_units_fixture()
        sys.exit(0)
    if a.only == "config":
        gen_config_fixture()
        sys.exit(0)
    if a.only == "label_prep":
        gen_label_prep_fixture()
        sys.exit(0)
    if a.only == "eval":
        gen_eval_fixture()
        sys.exit(0)
    if a.only == "pseudo_panoptic":
        gen_pseudo_panoptic_fixture()
        sys.exit(0)
    if a.only == "data":  # own process: the data stand-ins must be in place before detectron2.data is imported
        gen_data_fixture()
        sys.exit(0)
    if a.only in ("", "ops", "model", "model_small", "inference", "trajectory"):
        ra = import_reference()
        if a.only in ("", "ops"):
            gen_op_fixtures(ra)
        if a.only in ("", "model", "model_small"):
            gen_model_fixture("model_small", (192, 256), 2, 5)
        if a.only in ("", "trajectory"):
            gen_trajectory_fixture("trajectory_small", (192, 256), 2, 7)
        if a.only in ("", "inference"):
            gen_inference_fixture("inference_small", (192, 256), 2)
            gen_inference_fixture("inference_ragged", [(160, 224), (128, 192)], 2)
