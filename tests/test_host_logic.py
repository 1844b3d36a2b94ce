"""CPU tests of the host side: config tree, registries, model construction (names / counts pinned by the reference
fixture), structures, schedule, synthetic data, and that the C-ABI library exports what include/u2seg_hip.h declares."""
import ctypes
import json
import os
import zlib

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_DIR = os.path.join(ROOT, "configs", "COCO-PanopticSegmentation")


def _cfg(name="u2seg_R50_800.yaml", opts=()):
    from u2seg_amd.config import get_cfg

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(CFG_DIR, name))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"] + list(opts))
    return cfg


def test_config_chain_and_overrides():
    cfg = _cfg()
    assert cfg.MODEL.META_ARCHITECTURE == "PanopticFPN" and cfg.MODEL.ROI_HEADS.NAME == "CascadeROIHeads"
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 800 and cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES == 28
    assert cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN == 4000 and cfg.MODEL.RPN.NMS_THRESH == 0.65
    assert cfg.SOLVER.STEPS == (210000, 250000) and cfg.SOLVER.CLIP_GRADIENTS.ENABLED
    assert _cfg("u2seg_R50_300.yaml").MODEL.ROI_HEADS.NUM_CLASSES == 300
    assert _cfg("u2seg_eval_800.yaml").MODEL.ROI_HEADS.NUM_CLASSES == 800
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.SOLVER.BASE_LR = 1.0
    c2 = cfg.clone()
    c2.defrost()
    c2.merge_from_list(["SOLVER.BASE_LR", "0.5", "MODEL.RPN.IN_FEATURES", "['p2','p3']"])
    assert c2.SOLVER.BASE_LR == 0.5 and c2.MODEL.RPN.IN_FEATURES == ["p2", "p3"]
    with pytest.raises(KeyError):
        c2.merge_from_list(["SOLVER.NOT_A_KEY", 1])
    with pytest.raises(ValueError):
        c2.merge_from_list(["SOLVER.MAX_ITER", "abc"])
    assert "PanopticFPN" in cfg.dump()


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="reference tree not present")
def test_reference_yaml_chain_loads_unchanged():
    """The reference's own 3-file _BASE_ chain gives the same tree as this repo's flat files."""
    from u2seg_amd.config import get_cfg

    for name in ("u2seg_R50_800.yaml", "u2seg_R50_300.yaml", "u2seg_eval_800.yaml"):
        ref = get_cfg()
        ref.merge_from_file(os.path.join("/root/reference/configs/COCO-PanopticSegmentation", name))
        mine = _cfg(name)
        for k in ("WEIGHTS", "DEVICE"):
            mine.MODEL[k] = ref.MODEL[k]
        mine.OUTPUT_DIR = ref.OUTPUT_DIR
        assert mine.dump() == ref.dump(), name


def test_registries_and_model_tree():
    from u2seg_amd.modeling import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_BOX_HEAD_REGISTRY,
                                    ROI_HEADS_REGISTRY, ROI_MASK_HEAD_REGISTRY, RPN_HEAD_REGISTRY, SEM_SEG_HEADS_REGISTRY,
                                    ANCHOR_GENERATOR_REGISTRY, build_model)

    for reg, name in ((META_ARCH_REGISTRY, "PanopticFPN"), (BACKBONE_REGISTRY, "build_resnet_fpn_backbone"),
                      (PROPOSAL_GENERATOR_REGISTRY, "RPN"), (RPN_HEAD_REGISTRY, "StandardRPNHead"),
                      (ANCHOR_GENERATOR_REGISTRY, "DefaultAnchorGenerator"), (ROI_HEADS_REGISTRY, "CascadeROIHeads"),
                      (ROI_BOX_HEAD_REGISTRY, "FastRCNNConvFCHead"), (ROI_MASK_HEAD_REGISTRY, "MaskRCNNConvUpsampleHead"),
                      (SEM_SEG_HEADS_REGISTRY, "SemSegFPNHead")):
        assert reg.get(name) is not None
    with pytest.raises(KeyError):
        META_ARCH_REGISTRY.get("NoSuchArch")
    model = build_model(_cfg())
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "model_small.json")))
    sd = model.state_dict()
    assert sum(p.numel() for p in model.parameters()) == fx["num_params"] == 76066554
    assert len(sd) == fx["num_state_entries"] == 431
    # identical key names, in the identical order, as the reference model's state_dict
    assert zlib.crc32("\n".join(sd.keys()).encode()) == fx["state_dict_keys_crc32"]
    assert model.backbone.size_divisibility == 32 and model.sem_seg_head.ignore_value == 255
    assert sum(p.numel() for p in build_model(_cfg("u2seg_R50_300.yaml")).parameters()) == 74400554
    model.eval()
    assert not model.training


def test_structures():
    from u2seg_amd.structures import BitMasks, Boxes, ImageList, Instances, pairwise_iou

    b = Boxes(torch.tensor([[0.0, 0, 10, 10], [5, 5, 15, 15], [3, 3, 3, 9]]))
    assert b.area().tolist() == [100.0, 100.0, 0.0] and b.nonempty().tolist() == [True, True, False]
    iou = pairwise_iou(b[:2], b[:2])
    assert iou[0, 1] == pytest.approx(25 / 175)  # tests/structures/test_boxes.py-style known value
    b.clip((8, 12))
    assert b.tensor[1].tolist() == [5, 5, 12, 8]
    inst = Instances((20, 30), gt_boxes=Boxes(torch.zeros(3, 4)), gt_classes=torch.tensor([1, 2, 3]))
    assert len(inst[torch.tensor([True, False, True])]) == 2 and inst.image_size == (20, 30)
    with pytest.raises(AssertionError):
        inst.bad = torch.zeros(5)
    cat = Instances.cat([inst, inst])
    assert len(cat) == 6 and cat.gt_classes.tolist() == [1, 2, 3, 1, 2, 3]
    il = ImageList.from_tensors([torch.ones(3, 20, 31), torch.ones(3, 33, 17)], 32)
    assert tuple(il.tensor.shape) == (2, 3, 64, 32) and il.image_sizes == [(20, 31), (33, 17)]
    assert float(il.tensor[0, :, 20:, :].abs().sum()) == 0
    assert ImageList.padded_size([(800, 1333)], 32) == (800, 1344)
    bm = BitMasks(torch.zeros(2, 4, 4, dtype=torch.bool))
    assert len(bm) == 2 and bm.nonempty().tolist() == [False, False]


def test_anchor_generator_and_sampling():
    from u2seg_amd.modeling import subsample_labels
    from u2seg_amd.modeling.rpn import DefaultAnchorGenerator

    ag = DefaultAnchorGenerator(sizes=[[32, 64]], aspect_ratios=[[0.25, 1, 4]], strides=[4], offset=0.0)
    anc = ag.grid_anchors([(1, 2)])[0]
    # tests/modeling/test_anchor_generator.py:13-43
    expected = torch.tensor([[-32.0, -8.0, 32.0, 8.0], [-16.0, -16.0, 16.0, 16.0], [-8.0, -32.0, 8.0, 32.0],
                             [-64.0, -16.0, 64.0, 16.0], [-32.0, -32.0, 32.0, 32.0], [-16.0, -64.0, 16.0, 64.0],
                             [-28.0, -8.0, 36.0, 8.0], [-12.0, -16.0, 20.0, 16.0], [-4.0, -32.0, 12.0, 32.0],
                             [-60.0, -16.0, 68.0, 16.0], [-28.0, -32.0, 36.0, 32.0], [-12.0, -64.0, 20.0, 64.0]])
    assert torch.allclose(anc, expected)
    # tests/modeling/test_anchor_generator.py:45-72: the same cell anchors on a grid shifted by offset 0.5 (half a stride)
    shifted = DefaultAnchorGenerator(sizes=[[32, 64]], aspect_ratios=[[0.25, 1, 4]], strides=[4], offset=0.5).grid_anchors([(1, 2)])[0]
    assert torch.allclose(shifted, expected + 2.0)
    from oracle import ops as O

    cell = [O.generate_cell_anchors([32, 64], [0.25, 1, 4]).float()]
    assert torch.allclose(O.grid_anchors([(1, 2)], [4], cell, 0.5)[0], expected + 2.0)
    assert torch.allclose(O.grid_anchors([(1, 2)], [4], cell, 0.0)[0], expected)
    # tests/structures/test_imagelist.py:20-44: padding to a multiple of the divisibility, sizes remembered
    from u2seg_amd.structures import ImageList

    il = ImageList.from_tensors([torch.ones(3, 15, 20)], 4)
    assert tuple(il.tensor.shape) == (1, 3, 16, 20) and list(il.image_sizes[0]) == [15, 20]
    il = ImageList.from_tensors([torch.ones(3, 25, 20), torch.ones(3, 10, 10)], 4)
    assert tuple(il.tensor.shape) == (2, 3, 28, 20) and [list(x) for x in il.image_sizes] == [[25, 20], [10, 10]]
    assert float(il.tensor[1, :, 10:, :].abs().sum()) == 0 and float(il.tensor[1, :, :10, :10].sum()) == 300
    import numpy as np

    g = np.load(os.path.join(ROOT, "tests", "golden", "ops_golden.npz"))
    torch.manual_seed(11)
    from u2seg_amd.modeling import set_permutation_source

    set_permutation_source(lambda n, device=None: torch.randperm(n))
    pos, neg = subsample_labels(torch.from_numpy(g["sub_labels"]), 64, 0.25, 2)
    set_permutation_source(None)
    assert np.array_equal(pos.numpy(), g["sub_pos"]) and np.array_equal(neg.numpy(), g["sub_neg"])


def test_lr_schedule_and_synthetic_data():
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.solver import WarmupMultiStepLR

    class Opt:
        lr = 0.0

    s = WarmupMultiStepLR(Opt(), 0.01, [210000, 250000], 0.02, 0.001, 1000)
    assert s.get_lr(0) == pytest.approx(1e-5) and s.get_lr(500) == pytest.approx(0.5 * 1e-5 + 0.5 * 0.01)
    assert s.get_lr(1000) == pytest.approx(0.01) and s.get_lr(210000) == pytest.approx(2e-4)
    assert s.get_lr(260000) == pytest.approx(0.01 * 0.02 ** 2)
    a, b = make_synthetic_batch(2, height=64, width=96), make_synthetic_batch(2, height=64, width=96)
    assert torch.equal(a[1]["image"], b[1]["image"]) and a[0]["image"].dtype == torch.uint8
    x = a[0]
    assert x["sem_seg"].shape == (64, 96) and set(x["sem_seg"].unique().tolist()) <= set(range(28)) | {255}
    inst = x["instances"]
    assert 3 <= len(inst) <= 12 and inst.gt_masks.tensor.shape[1:] == (64, 96) and int(inst.gt_classes.max()) < 800


def test_cabi_library_exports_every_declared_symbol():
    from u2seg_amd import _hip

    decl = _hip.declared_symbols()
    assert len(decl) >= 36 and "u2_conv_igemm" in decl and "u2_kmeans_assign" in decl
    assert os.path.exists(_hip.lib_path()), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_hip.lib_path())
    for name in decl:
        assert hasattr(lib, name), name
    lib.u2_abi_version.restype = ctypes.c_int
    assert lib.u2_abi_version() == 1
    assert _hip.call_nostream("u2_nms_workspace_bytes", 2, 130) == 2 * 130 * 3 * 8


def test_hot_path_fails_loudly_without_gpu():
    """No CPU fallback: a HIP function on CPU tensors is an error, not a silent ATen path."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from u2seg_amd.layers import functional as F

    with pytest.raises((AssertionError, RuntimeError)):
        F.conv2d(torch.zeros((1, 4, 4, 32), dtype=torch.bfloat16), torch.zeros(32, 32, 1, 1))


def test_lazy_bitmasks_and_batched_views():
    """Host-side structures behind the batched bookkeeping: lazy BitMasks selection (composition of index vectors, no
    gather until .tensor is read), the lazy proposal list, padded targets and the device-constant cache."""
    from u2seg_amd.modeling.batched import BatchList, LazyProposals, PaddedTargets, device_constant, image_index, proposals_from_list
    from u2seg_amd.structures import BitMasks, Boxes, Instances

    g = torch.Generator().manual_seed(0)
    base = torch.rand((5, 6, 7), generator=g) > 0.5
    bm = BitMasks(base)
    sel = bm[torch.tensor([4, 1, 1, 0])]
    assert len(sel) == 4 and sel._base is bm._base  # nothing gathered yet
    sub = sel[torch.tensor([True, False, True, True])]
    assert len(sub) == 3 and torch.equal(sub.tensor, base[torch.tensor([4, 1, 0])])
    assert torch.equal(sel[2:3].tensor if False else sel.tensor[2:3], base[1:2])
    assert torch.equal(BitMasks.cat([sel, sub]).tensor, torch.cat([base[torch.tensor([4, 1, 1, 0])], base[torch.tensor([4, 1, 0])]]))

    # LazyProposals: padded tensors + counts -> ragged list[Instances] on demand; FloatingPointError on non-finite flag
    boxes = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 3, 4)
    logits = torch.arange(6, dtype=torch.float32).view(2, 3)
    lp = LazyProposals([(10, 12), (8, 9)], boxes, logits, torch.tensor([2, 3], dtype=torch.int32), torch.tensor(True), True)
    assert len(lp) == 2 and lp._items is None
    assert [len(p) for p in lp] == [2, 3] and lp[1].image_size == (8, 9)
    assert torch.equal(lp[0].proposal_boxes.tensor, boxes[0, :2]) and torch.equal(lp[1].objectness_logits, logits[1])
    bad = LazyProposals([(10, 12)], boxes[:1], logits[:1], torch.tensor([1], dtype=torch.int32), torch.tensor(False), True)
    with pytest.raises(FloatingPointError):
        bad[0]
    back = proposals_from_list(list(lp))
    assert back.counts.tolist() == [2, 3] and torch.equal(back.boxes[1], boxes[1]) and float(back.boxes[0, 2].abs().sum()) == 0

    # PaddedTargets: zero padded, cached per gt list
    t0, t1 = Instances((10, 12)), Instances((8, 9))
    t0.gt_boxes, t0.gt_classes = Boxes(torch.tensor([[1.0, 2, 3, 4]])), torch.tensor([7])
    t1.gt_boxes, t1.gt_classes = Boxes(torch.zeros((0, 4))), torch.zeros(0, dtype=torch.int64)
    pt = PaddedTargets.of([t0, t1], "cpu")
    assert pt.boxes.shape == (2, 1, 4) and pt.counts.tolist() == [1, 0] and pt.classes.tolist() == [[7], [0]]
    assert PaddedTargets.of([t0, t1], "cpu") is pt

    assert device_constant([1, 2, 3], torch.int32, "cpu") is device_constant([1, 2, 3], torch.int32, "cpu")
    assert image_index([2, 0, 3], "cpu").tolist() == [0, 0, 2, 2, 2]
    bl = BatchList([t0])
    assert not bl.stacked and len(bl) == 1


def test_batchnorm_counts_batches_lazily():
    from u2seg_amd.layers.modules import BatchNorm2d

    bn = BatchNorm2d(8)
    for _ in range(3):
        bn.count_batch()
    assert int(bn.num_batches_tracked) == 0  # no per-step device op ...
    assert int(bn.state_dict()["num_batches_tracked"]) == 3  # ... folded in when the buffer is read
    bn.load_state_dict(bn.state_dict())
    bn.count_batch()
    assert int(bn.state_dict()["num_batches_tracked"]) == 4


def test_detection_checkpointer_formats(tmp_path):
    """checkpoint/detection_checkpoint.py:70-143: Detectron2-zoo .pkl (ndarray dict, suffix matching heuristics, shape
    mismatches skipped and reported) and torch .pth round trips into the reference's state-dict names."""
    import pickle

    from u2seg_amd.checkpoint import DetectionCheckpointer
    from u2seg_amd.modeling import build_model

    torch.manual_seed(0)
    model = build_model(_cfg())
    ref = {k: v.clone() for k, v in model.state_dict().items()}
    # 1. a backbone-only zoo file whose names lack the 'backbone.bottom_up.' prefix, one tensor with a wrong shape
    prefix = "backbone.bottom_up."
    blob = {k[len(prefix):]: (v.numpy() + 1.0) for k, v in ref.items() if k.startswith(prefix) and v.dtype == torch.float32}
    bad = "res2.0.conv1.weight"
    blob[bad] = np.zeros((3, 3), dtype=np.float32)
    p = tmp_path / "dino_like.pkl"
    with open(p, "wb") as f:
        pickle.dump({"model": blob, "__author__": "test", "matching_heuristics": True}, f)
    ck = DetectionCheckpointer(model)
    rest = ck.load(str(p))
    assert "model" not in rest and rest.get("__author__") == "test"
    inc = ck.last_incompatible
    assert [x[0] for x in inc.incorrect_shapes] == [prefix + bad]
    assert any(k.startswith("roi_heads.") for k in inc.missing_keys) and not inc.unexpected_keys
    sd = model.state_dict()
    assert torch.equal(sd[prefix + "res3.1.conv2.weight"], ref[prefix + "res3.1.conv2.weight"] + 1.0)
    assert torch.equal(sd[prefix + bad], ref[prefix + bad])  # skipped
    assert torch.equal(sd["roi_heads.mask_head.predictor.weight"], ref["roi_heads.mask_head.predictor.weight"])
    # 2. full .pth round trip (with an extra item, as the reference's periodic checkpointer writes)
    ck2 = DetectionCheckpointer(model, str(tmp_path))
    path = ck2.save("model_0000009", iteration=9)
    with torch.no_grad():
        for v in model.parameters():
            v.zero_()
    rest = DetectionCheckpointer(model, str(tmp_path)).resume_or_load("", resume=True)
    assert rest["iteration"] == 9 and os.path.basename(path) == "model_0000009.pth"
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), sd.values()))


def test_checkpoint_name_matching_equals_reference():
    """checkpoint/c2_model_loading.py:209-330: the suffix-matching heuristic that maps a prefix-free backbone file (the format of
    U2Seg's dino_RN50_pretrain_d2_format.pkl, u2seg_R50_800.yaml:6) onto the model's names.  The fixture is the output of the
    reference's own align_and_update_state_dicts on the reference model's 431 keys (make_fixtures.py --only checkpoint);
    u2seg_amd.checkpoint.align_by_suffix must route every checkpoint tensor to the same model key, and this package's model
    must expose exactly those keys and shapes."""
    import json

    from u2seg_amd.checkpoint import align_by_suffix
    from u2seg_amd.modeling import build_model

    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "checkpoint_matching_golden.json")))
    model_sd = build_model(_cfg()).state_dict()
    assert {k: list(v.shape) for k, v in model_sd.items()} == fx["model_keys"]
    ckpt = {k: torch.empty(shape, device="meta") for k, shape in fx["ckpt_shapes"].items()}
    mine = align_by_suffix(list(model_sd.keys()), ckpt)
    mine_map = {mk: next(ck for ck, t in ckpt.items() if t is v) for mk, v in mine.items()}
    wrong_shape = "res2.0.conv1.weight"  # the reference drops a shape mismatch while matching, this package right after it
    want = dict(fx["result"])
    assert want.pop(wrong_shape) == wrong_shape and mine_map.pop("backbone.bottom_up." + wrong_shape) == wrong_shape
    assert mine_map == want
    assert sum(1 for a, b in want.items() if a != b) == 264 and want["stem.fc.weight"] == "stem.fc.weight"  # unused: passed through


def test_reference_checkpoint_files_load_through_the_loader():
    """files in the reference's two on-disk forms, written from the REFERENCE's model / optimizer /
    scheduler objects (make_fixtures.py --only checkpoint_files; u2seg_R50_800 with its widths reduced through its own config
    keys), through u2seg_amd.checkpoint.DetectionCheckpointer (checkpoint/detection_checkpoint.py:70-143):
      * the `.pth` form fvcore's Checkpointer writes - every one of the 431 tensors arrives bit-identical under its own name,
        nothing is missing or unexpected, optimizer / scheduler / iteration come back untouched (load without --resume);
      * the Detectron2 model-zoo `.pkl` form (prefix-free backbone arrays + matching_heuristics) - every array lands on the
        model key the reference's align_and_update_state_dicts assigns it to, the classifier left-over is reported and not
        loaded, everything outside the backbone keeps its initial value."""
    import json
    import pickle

    from u2seg_amd.checkpoint import DetectionCheckpointer
    from u2seg_amd.modeling import build_model

    gdir = os.path.join(ROOT, "tests", "golden")
    fx = json.load(open(os.path.join(gdir, "checkpoint_files_golden.json")))
    cfg = _cfg()
    cfg.merge_from_list(fx["opts"])
    torch.manual_seed(3)
    model = build_model(cfg)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == fx["pth_shapes"]
    ck = DetectionCheckpointer(model)
    rest = ck.resume_or_load(os.path.join(gdir, "checkpoint_small.pth"), resume=False)
    inc = ck.last_incompatible
    assert not inc.missing_keys and not inc.unexpected_keys and not inc.incorrect_shapes
    assert rest["iteration"] == fx["iteration"] and "optimizer" in rest and "scheduler" in rest
    assert rest["optimizer"]["param_groups"] and "last_epoch" in rest["scheduler"]  # the reference objects' own state dicts
    got = {k: zlib.crc32(v.contiguous().numpy().tobytes()) for k, v in model.state_dict().items()}
    assert got == fx["pth_crc32"]

    # the model-zoo pickle on a freshly initialised model
    torch.manual_seed(4)
    model = build_model(cfg)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    ck = DetectionCheckpointer(model)
    ck.load(os.path.join(gdir, "checkpoint_small_d2.pkl"), checkpointables=[])
    zoo = pickle.load(open(os.path.join(gdir, "checkpoint_small_d2.pkl"), "rb"))["model"]
    assign = fx["pkl_assignment"]
    after = model.state_dict()
    loaded = 0
    for mk, src in assign.items():
        if mk == src:          # passed through unmatched by the reference (the classifier left-over): must not be in the model
            assert mk not in after
            continue
        assert torch.equal(after[mk], torch.from_numpy(zoo[src])), mk
        loaded += 1
    assert loaded == 265 and ck.last_incompatible.unexpected_keys == ["stem.fc.weight"]
    untouched = [k for k in after if k not in assign]
    assert untouched and all(torch.equal(after[k], before[k]) for k in untouched)
    assert sorted(k for k in ck.last_incompatible.missing_keys) == sorted(untouched)


def test_reference_optimizer_and_scheduler_state_round_trip(tmp_path):
    """SURVEY 8(f) row 2, --resume: tests/golden/checkpoint_resume.pth was written by the REFERENCE after two steps of a run
    (make_fixtures.py --only resume: its model, its clip-wrapped torch.optim.SGD, its WarmupMultiStepLR, in the nesting its
    DefaultTrainer writes: checkpoint/detection_checkpoint.py:70-143, engine/defaults.py:389-421,499-506).  Through
    DetectionCheckpointer(model, optimizer=FlatSGD, scheduler=...):
      * FlatSGD forms the reference's parameter groups (reduce_param_groups, solver/build.py:255-279) and torch's numbering;
      * every weight and every momentum buffer arrives bit-identical under its parameter's NAME, lr / iteration / schedule
        position are the file's;
      * FlatSGD.state_dict() is the dict torch.optim.SGD wrote (same group entries, same tensors) and a real torch.optim.SGD
        over the same grouping loads it; a file written by this package's checkpointer carries the reference's nesting too and
        reads back into a fresh optimizer unchanged."""
    from u2seg_amd.checkpoint import DetectionCheckpointer
    from u2seg_amd.modeling import build_model
    from u2seg_amd.solver import build_lr_scheduler, build_optimizer

    gdir = os.path.join(ROOT, "tests", "golden")
    fx = json.load(open(os.path.join(gdir, "resume_golden.json")))
    path = os.path.join(gdir, "checkpoint_resume.pth")
    crc = lambda t: zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())

    def fresh(seed):
        torch.manual_seed(seed)
        cfg = _cfg(opts=fx["opts"])
        model = build_model(cfg)
        opt = build_optimizer(cfg, model)
        return cfg, model, opt, build_lr_scheduler(cfg, opt)

    cfg, model, opt, sched = fresh(5)
    names = {id(p): k for k, p in model.named_parameters()}
    assert [len(m) for m in opt.group_members] == fx["group_sizes"] and opt.group_wd == fx["group_weight_decay"]
    numbering = [names[id(opt.params[i])] for members in opt.group_members for i in members]
    assert numbering == [fx["numbering"][str(i)] for i in range(len(numbering))]

    ck = DetectionCheckpointer(model, optimizer=opt, scheduler=sched)
    rest = ck.load(path)
    assert rest["iteration"] == fx["saved_iteration"] and "optimizer" not in rest and "scheduler" not in rest
    assert not ck.last_incompatible.missing_keys and not ck.last_incompatible.unexpected_keys
    assert {k: crc(v) for k, v in model.state_dict().items()} == fx["model_crc32"]
    mom = {names[id(p)]: crc(opt.flat_mom[off : off + p.numel()]) for p, off in zip(opt.params, opt.param_offset)}
    assert mom == fx["momentum_crc32"]
    nxt = fx["saved_iteration"] + 1
    assert opt.lr == fx["lr"][nxt] and sched.last_iter == nxt and sched.get_lr(nxt) == pytest.approx(fx["lr"][nxt], rel=1e-12)
    # parameters still alias the arena after load_state_dict (the kernels and the optimizer must see the loaded values)
    assert all(p.data_ptr() == opt.flat_param.data_ptr() + 4 * off for p, off in zip(opt.params, opt.param_offset))

    ref = torch.load(path, weights_only=False, map_location="cpu")["trainer"]
    ref_opt, ref_sched = ref["_trainer"]["optimizer"], ref["hooks"]["LRScheduler"]
    mine = opt.state_dict()
    assert len(mine["param_groups"]) == len(ref_opt["param_groups"])
    for a, b in zip(mine["param_groups"], ref_opt["param_groups"]):
        assert a == b, (a, b)
    assert sorted(mine["state"]) == sorted(ref_opt["state"])
    assert all(torch.equal(mine["state"][i]["momentum_buffer"], ref_opt["state"][i]["momentum_buffer"]) for i in mine["state"])
    ssd = sched.state_dict()
    assert ssd["last_epoch"] == ref_sched["last_epoch"] and ssd["base_lrs"] == ref_sched["base_lrs"]
    # torch's own optimizer accepts what FlatSGD wrote
    clones = [torch.nn.Parameter(p.detach().clone()) for p in opt.params]
    tsgd = torch.optim.SGD([{"params": [clones[i] for i in members], "weight_decay": wd}
                            for wd, members in zip(opt.group_wd, opt.group_members)], lr=1.0, momentum=0.5)
    tsgd.load_state_dict(mine)
    assert tsgd.param_groups[0]["lr"] == opt.lr and tsgd.param_groups[0]["momentum"] == 0.9
    assert all(torch.equal(tsgd.state[clones[i]]["momentum_buffer"].reshape(-1), opt.flat_mom[off : off + clones[i].numel()])
               for i, off in enumerate(opt.param_offset))

    # written by this package, read by this package: same state; the file has the reference's nesting as well
    out = DetectionCheckpointer(model, str(tmp_path), optimizer=opt, scheduler=sched).save("model_0000001", iteration=1)
    back = torch.load(out, weights_only=False, map_location="cpu")
    assert back["trainer"]["iteration"] == 1 and back["trainer"]["hooks"]["LRScheduler"]["last_epoch"] == nxt
    assert back["trainer"]["_trainer"]["optimizer"]["param_groups"] == mine["param_groups"]
    cfg2, model2, opt2, sched2 = fresh(6)
    ck2 = DetectionCheckpointer(model2, str(tmp_path), optimizer=opt2, scheduler=sched2)
    assert ck2.has_checkpoint()
    rest2 = ck2.resume_or_load("", resume=True)
    assert rest2["iteration"] == 1 and torch.equal(opt2.flat_mom, opt.flat_mom) and torch.equal(opt2.flat_param, opt.flat_param)
    assert opt2.lr == opt.lr and sched2.last_iter == nxt
    # a plain_train_net.py-style file (top-level "optimizer" / "scheduler") loads the same way; weights only without --resume
    torch.save({"model": back["model"], "optimizer": ref_opt, "scheduler": ref_sched, "iteration": 1}, str(tmp_path / "plain.pth"))
    cfg3, model3, opt3, sched3 = fresh(7)
    DetectionCheckpointer(model3, optimizer=opt3, scheduler=sched3).load(str(tmp_path / "plain.pth"))
    assert torch.equal(opt3.flat_mom, opt.flat_mom) and opt3.lr == opt.lr and sched3.last_iter == nxt
    cfg4, model4, opt4, sched4 = fresh(8)
    rest4 = DetectionCheckpointer(model4, optimizer=opt4, scheduler=sched4).resume_or_load(path, resume=False)
    assert torch.equal(opt4.flat_param, opt.flat_param) and float(opt4.flat_mom.abs().sum()) == 0.0 and sched4.last_iter == 0
    assert "optimizer" in rest4 and rest4["iteration"] == fx["saved_iteration"]


def test_fullwidth_reference_checkpoint_host_side(tmp_path):
    """f2 at FULL width (RES2_OUT_CHANNELS 256, 800 + 1 classes, 76 066 554 parameters): the checkpoint
    content the reference's own model / optimizer / scheduler objects produce (tests/golden/fullwidth_checkpoint_golden.json: crc32
    of all 431 tensors and 248 momentum buffers after one SGD step, make_fixtures.py --only fullwidth) is rebuilt with torch
    alone - equal by crc32 - saved in the reference's nesting and loaded through DetectionCheckpointer(model, optimizer=FlatSGD,
    scheduler=...): every weight and momentum buffer arrives bit for bit under its NAME, groups / numbering / lr / schedule
    position are the file's, and what FlatSGD writes back is the reference's param_groups.  (The same file on the device:
    tests/test_gpu_bookkeeping.py::test_fullwidth_reference_checkpoint_on_the_device.)"""
    from tests.parity_checks import write_fullwidth_reference_checkpoint
    from u2seg_amd.checkpoint import DetectionCheckpointer
    from u2seg_amd.modeling import build_model
    from u2seg_amd.solver import build_lr_scheduler, build_optimizer

    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullwidth_checkpoint_golden.json")))
    cfg = _cfg()
    torch.manual_seed(3)
    model = build_model(cfg)
    assert sum(p.numel() for p in model.parameters()) == fx["num_parameters"] == 76066554
    opt = build_optimizer(cfg, model)
    sched = build_lr_scheduler(cfg, opt)
    names = {id(p): k for k, p in model.named_parameters()}
    assert [names[id(opt.params[i])] for m in opt.group_members for i in m] == fx["numbering"]
    assert [len(m) for m in opt.group_members] == [g["n"] for g in fx["param_groups"]]
    path = str(tmp_path / "fullwidth.pth")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    state, mom = write_fullwidth_reference_checkpoint(shapes, [k for k, _ in model.named_parameters()], fx, path)
    ck = DetectionCheckpointer(model, optimizer=opt, scheduler=sched)
    rest = ck.load(path)
    assert not ck.last_incompatible.missing_keys and not ck.last_incompatible.unexpected_keys and rest["iteration"] == 0
    crc = lambda t: zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())
    assert {k: crc(v) for k, v in model.state_dict().items()} == fx["model_crc32"]
    got = {names[id(p)]: crc(opt.flat_mom[off : off + p.numel()]) for p, off in zip(opt.params, opt.param_offset)}
    assert got == fx["momentum_crc32"]
    # (the fixture's scheduler is the reference's plain WarmupMultiStepLR class, lr = base * (f (1 - a) + a); this package follows
    #  build_lr_scheduler's fvcore composite, start (1 - a) + end a: the same number to an ulp)
    assert opt.lr == pytest.approx(fx["param_groups"][0]["lr"], rel=1e-14) and opt.base_lr == fx["param_groups"][0]["initial_lr"]
    assert sched.last_iter == fx["scheduler"]["last_epoch"] and sched.get_lr(sched.last_iter) == opt.lr
    mine = opt.state_dict()["param_groups"]
    for a, b in zip(mine, fx["param_groups"]):
        a, b = {k: v for k, v in a.items() if k != "params"}, {k: v for k, v in b.items() if k != "n"}
        assert a.pop("lr") == pytest.approx(b.pop("lr"), rel=1e-14) and a == b, (a, b)
    assert all(p.data_ptr() == opt.flat_param.data_ptr() + 4 * off for p, off in zip(opt.params, opt.param_offset))


def test_param_groups_follow_the_reference_override_dicts():
    """The reference merges parameters by their override DICT (solver/build.py:123-129,181-236,255-279): a norm / bias override
    that equals WEIGHT_DECAY still forms a group of its own, and all overridden parameters of one value share one.  Fixture:
    make_fixtures.py --only param_groups (the reference's build_optimizer on the reduced u2seg_R50_800, eight triples).  The
    numbering decides whether a reference checkpoint's optimizer state loads at all."""
    from u2seg_amd.modeling import build_model
    from u2seg_amd.solver import build_optimizer

    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "param_groups_golden.json")))
    cfg = _cfg(opts=fx["opts"])
    model = build_model(cfg)
    names = {id(p): k for k, p in model.named_parameters()}
    for case in fx["cases"]:
        cfg.SOLVER.WEIGHT_DECAY = case["weight_decay"]
        cfg.SOLVER.WEIGHT_DECAY_NORM = case["weight_decay_norm"]
        cfg.SOLVER.WEIGHT_DECAY_BIAS = case["weight_decay_bias"]
        state = {k: v.clone() for k, v in model.state_dict().items()}
        opt = build_optimizer(cfg, model)
        assert [len(m) for m in opt.group_members] == case["group_sizes"], case
        assert opt.group_wd == case["group_weight_decay"], case
        assert [names[id(opt.params[m[0]])] for m in opt.group_members] == case["first_names"]
        order = [names[id(opt.params[i])] for m in opt.group_members for i in m]
        assert zlib.crc32("\n".join(order).encode()) == case["numbering_crc32"]
        # per-parameter decay the kernel applies = the group's value
        wd = opt.wd.tolist()
        assert all(wd[i] == pytest.approx(g) for g, m in zip(opt.group_wd, opt.group_members) for i in m)
        # what it writes loads into torch's SGD over the same grouping, and a group's decay is addressed by POSITION on load
        sd = opt.state_dict()
        clones = [torch.nn.Parameter(p.detach().clone()) for p in opt.params]
        tsgd = torch.optim.SGD([{"params": [clones[i] for i in m], "weight_decay": g}
                                for g, m in zip(opt.group_wd, opt.group_members)], lr=1.0, momentum=0.9)
        tsgd.load_state_dict(sd)
        if len(opt.group_wd) >= 2:
            sd["param_groups"][-1]["weight_decay"] = 0.125
            for g in sd["param_groups"]:
                g["initial_lr"] = 0.5
            opt.load_state_dict(sd)
            assert opt.group_wd[-1] == 0.125 and opt.group_wd[:-1] == case["group_weight_decay"][:-1] and opt.base_lr == 0.5
            wd = opt.wd.tolist()
            assert all(wd[i] == pytest.approx(g) for g, m in zip(opt.group_wd, opt.group_members) for i in m)
        model.load_state_dict(state)


def test_resolved_configs_equal_reference():
    """Every key this package's config tree holds has the value the reference resolves for the same yaml (its defaults.py +
    _BASE_ chain; fixture: tests/golden/make_fixtures.py --only config), for the four U2Seg train / eval configs - both
    through this repo's flat copies and the values the model, solver and data path read."""
    import json

    from u2seg_amd.config import get_cfg

    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "config_golden.json")))

    def norm(x):
        if isinstance(x, dict):
            return {k: norm(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [norm(v) for v in x]
        return x

    def compare(mine, ref, path, skipped):
        for k, v in mine.items():
            here = path + [k]
            if k not in ref:
                skipped.append(".".join(here))
                continue
            if isinstance(v, dict):
                assert isinstance(ref[k], dict), here
                compare(v, ref[k], here, skipped)
            elif ".".join(here) == "MODEL.WEIGHTS" and str(ref[k]).startswith("/home/"):
                assert v == ""  # the training yamls point into the author's home directory; here: empty = random init
            else:
                assert norm(v) == ref[k], (".".join(here), v, ref[k])

    for name, ref in gold.items():
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", name + ".yaml"))
        skipped = []
        compare(norm(cfg), ref, [], skipped)
        assert skipped == [], skipped  # nothing in this tree is unknown to the reference
        for key in ("MODEL", "SOLVER", "INPUT", "DATASETS", "DATALOADER", "TEST"):
            assert key in cfg
    # spot values the hot path depends on
    r = gold["u2seg_R50_800"]
    assert r["MODEL"]["ROI_HEADS"]["NUM_CLASSES"] == 800 and r["SOLVER"]["GAMMA"] == 0.02
    assert r["DATALOADER"]["FILTER_EMPTY_ANNOTATIONS"] is False and r["INPUT"]["MASK_FORMAT"] == "bitmask"


def test_lr_schedule_resume_matches_uninterrupted_run():
    """A run resumed at iteration k continues the reference's schedule (warm-up and STEPS milestones counted from iteration 0,
    solver/build.py:283-323 + the checkpointer restoring the scheduler): lr(k), lr(k+1), ... equal an uninterrupted run's."""
    from u2seg_amd.solver.build import WarmupMultiStepLR

    class Opt:
        lr = 0.0

    def make():
        o = Opt()
        return o, WarmupMultiStepLR(o, 0.01, [30, 50], 0.02, 1e-3, 20)

    full_opt, full = make()
    lrs = []
    for _ in range(70):
        lrs.append(full_opt.lr)
        full.step()
    for k in (7, 20, 31, 55):  # inside the warm-up, at its end, after each milestone
        o, s = make()
        s.resume_at(k)
        for it in range(k, 70):
            assert o.lr == lrs[it], (k, it)
            s.step()
        o2, s2 = make()
        s2.load_state_dict({"last_iter": k})
        assert o2.lr == lrs[k]
    assert lrs[0] == pytest.approx(1e-5) and lrs[20] == 0.01 and lrs[30] == pytest.approx(2e-4) and lrs[50] == pytest.approx(4e-6)


def test_fan_out_handles_sum_gradients_on_cpu():
    """functional.fan_out / PanopticFPN._fan_out_features: k autograd handles on one tensor, one handle per reader of a map that
    several consumers share, the tensor itself for a single reader; the gradients of all handles add up (CPU: torch adds)."""
    import torch

    from u2seg_amd.layers import functional as F
    from u2seg_amd.modeling.panoptic_fpn import GeneralizedRCNN

    x = torch.randn(3, 4, requires_grad=True)
    a, b, c = F.fan_out(x, 3)
    assert a.data_ptr() == x.data_ptr() and b.data_ptr() == x.data_ptr()
    (a * 1.0 + b * 2.0 + c * 3.0).sum().backward()
    assert torch.allclose(x.grad, torch.full_like(x, 6.0))
    with torch.no_grad():
        assert all(h is x for h in F.fan_out(x, 3))  # no autograd: the tensor itself
    feats = {"p2": torch.randn(2, 2, requires_grad=True), "p3": torch.randn(2, 2, requires_grad=True),
             "p6": torch.randn(2, 2, requires_grad=True)}
    sem, rpn, roi = GeneralizedRCNN._fan_out_features(feats, [["p2", "p3"], ["p2", "p3", "p6"], ["p2"]])
    assert rpn["p6"] is feats["p6"] and sem["p6"] is feats["p6"]          # one reader: untouched
    assert sem["p2"] is not feats["p2"] and rpn["p2"] is not sem["p2"]    # three readers: three handles
    (sem["p2"].sum() + 2 * rpn["p2"].sum() + 4 * roi["p2"].sum() + sem["p3"].sum() + rpn["p3"].sum()).backward()
    assert torch.allclose(feats["p2"].grad, torch.full((2, 2), 7.0)) and torch.allclose(feats["p3"].grad, torch.full((2, 2), 2.0))


def test_topk_segment_chooser():
    """layers/functional.py:_topk_segments - long rows are cut into equal segments only when they tile the row exactly, stay
    several times longer than k and do not exceed the chip's work-group slots."""
    from u2seg_amd.layers.functional import _topk_segments

    assert _topk_segments(16, 1000, 1, 1, 1000, 100) == 1                         # short rows: one pass
    s = _topk_segments(16, 201600, 3, 32, 67200 * 32, 2000)                       # stride-4 anchors through the (3, 32) view
    assert s > 1 and 67200 % s == 0 and 16 * s <= 256 and 201600 // s >= 4096
    s = _topk_segments(16, 268569, 1, 1, 268569, 256)                             # all anchors: 268 569 = 3^3 * 7^3 * 29
    assert s in (3, 7, 9) and 268569 % s == 0
    assert _topk_segments(16, 65537, 1, 1, 65537, 100) == 1                       # a prime length cannot be tiled
    assert _topk_segments(200, 201600, 3, 32, 67200 * 32, 2000) == 1              # many rows already fill the chip
    assert _topk_segments(16, 201600, 3, 32, 67200 * 32 + 8, 2000) == 1           # rows with a gap behind them: not a plain view


def test_eval_fold_cache_and_block_flags():
    """Conv2d._folded_eval (inference: fixed-statistics norm folded into the conv) is cached until one of its tensors changes
    through torch; ResNet marks the first block of every stage (its shortcut buffer is not given up at inference) and the last
    block of a stage that is also a backbone output (third gradient handle)."""
    import torch

    from u2seg_amd.config import get_cfg
    from u2seg_amd.layers.modules import Conv2d, get_norm
    from u2seg_amd.modeling import build_model

    conv = Conv2d(8, 4, 1, bias=False, norm=get_norm("BN", 4)).eval()
    with torch.no_grad():
        conv.norm.running_var.fill_(3.0)
        conv.norm.running_mean.fill_(0.5)
        w1, b1, sc1, sh1 = conv._folded_eval()
        assert conv._folded_eval()[0] is w1                                        # cached
        scale = conv.norm.weight / torch.sqrt(conv.norm.running_var + conv.norm.eps)
        assert torch.allclose(w1, conv.weight * scale.view(-1, 1, 1, 1)) and torch.allclose(b1, conv.norm.bias - 0.5 * scale)
        conv.norm.running_mean.add_(1.0)                                           # in-place change: version counter moves
        w2, b2, _, _ = conv._folded_eval()
        assert w2 is not w1 and torch.allclose(b2, conv.norm.bias - 1.5 * scale)
    cfg = get_cfg()
    cfg.merge_from_file("configs/COCO-PanopticSegmentation/u2seg_R50_800.yaml")
    cfg.MODEL.DEVICE = "cpu"
    bottom_up = build_model(cfg).backbone.bottom_up
    for name, stage in zip(bottom_up.stage_names, bottom_up.stages):
        blocks = list(stage)
        assert blocks[0].first_in_stage and not any(b.first_in_stage for b in blocks[1:])
        assert not any(b.third_handle for b in blocks[:-1])
        assert blocks[-1].third_handle == (name != bottom_up.stage_names[-1])      # res2-res4 feed the next stage AND the FPN


def test_detector_postprocess_batch_equals_per_image_rule():
    """detector_postprocess_batch rescales, clips and filters the boxes of all images with one set of tensor ops; the result
    must be what postprocessing.py:9-74 does image by image (Boxes.scale, Boxes.clip, nonempty), including images with no
    boxes, boxes that become empty after clipping and different output sizes per image."""
    from u2seg_amd.modeling.inference import detector_postprocess_batch
    from u2seg_amd.structures import Boxes, Instances

    g = torch.Generator().manual_seed(0)
    res, sizes = [], []
    for i, n in enumerate([7, 0, 3, 12, 1]):
        inst = Instances((480 + 10 * i, 640))
        b = torch.rand((n, 4), generator=g) * 700 - 30
        b[:, 2:] = b[:, :2] + torch.rand((n, 2), generator=g) * 200 - 20  # some with negative extent
        inst.pred_boxes = Boxes(b)
        inst.scores = torch.rand(n, generator=g)
        inst.pred_classes = torch.randint(0, 5, (n,), generator=g)
        res.append(inst)
        sizes.append((400 + 7 * i, 500 + 3 * i))
    out = detector_postprocess_batch(res, sizes)
    dropped = 0
    for inst, (oh, ow), o in zip(res, sizes, out):
        sx, sy = ow / inst.image_size[1], oh / inst.image_size[0]
        t = inst.pred_boxes.tensor * torch.tensor([sx, sy, sx, sy])
        t = torch.stack((t[:, 0].clamp(0, ow), t[:, 1].clamp(0, oh), t[:, 2].clamp(0, ow), t[:, 3].clamp(0, oh)), -1)
        keep = ((t[:, 2] - t[:, 0]) > 0) & ((t[:, 3] - t[:, 1]) > 0)
        dropped += int((~keep).sum())
        assert torch.equal(o.pred_boxes.tensor, t[keep]) and torch.equal(o.scores, inst.scores[keep])
        assert torch.equal(o.pred_classes, inst.pred_classes[keep]) and o.image_size == (oh, ow)
    assert dropped > 0


def test_topk_rows_sorted_fallback_total_order():
    """Selections with k beyond the HIP kernel's LDS-resident limit (16 384; e.g. the reference's default PRE_NMS_TOPK of 12 000
    summed over the levels in the score sort before NMS) go through a stable device sort: same total order (value, then index),
    same outputs as the kernel's contract - here against an explicit per-row stable sort, with ties, a mask, +-0 and the
    (group, pitch) view of a 32-wide map."""
    import u2seg_amd.layers.functional as F

    g = torch.Generator().manual_seed(1)
    rows, n, k = 3, 40000, 20000
    v = (torch.randn((rows, n), generator=g) * 2).mul(4).round().div(4)
    v[0, 5], v[0, 6] = -0.0, 0.0
    mask = torch.randint(0, 2, (rows, n), generator=g).to(torch.int8)
    mask[2] = 0
    mask[2, :100] = 1  # fewer participants than k
    for largest in (True, False):
        out, idx, cnt = F._topk_rows_sorted(dict(vals=v, k=k, largest=largest, mask=mask, mask_value=1))
        for r in range(rows):
            sel = torch.nonzero(mask[r] == 1)[:, 0]
            ref = sel[torch.sort(v[r, sel], descending=largest, stable=True)[1]][:k]
            c = min(k, len(sel))
            assert int(cnt[r]) == c and torch.equal(idx[r, :c].long(), ref[:c]) and torch.equal(out[r, :c], v[r, ref[:c]])
            assert bool((idx[r, c:] == 0).all()) and bool(torch.isinf(out[r, c:]).all())
    m = torch.randn((2, 5000, 32), generator=g).bfloat16()
    out, idx, cnt = F._topk_rows_sorted(dict(vals=m, k=14000, largest=True, group=3, pitch=32, n=15000))
    flat = m[..., :3].reshape(2, -1).float()
    for r in range(2):
        assert torch.equal(idx[r].long(), torch.sort(flat[r], descending=True, stable=True)[1][:14000])
    assert F._TOPK_MAX_K == 16384


def test_paste_reach_rectangle_contains_every_pixel_the_exact_condition_accepts():
    """postprocess.hip paste_reach: the rectangle the interior pass of u2_paste_masks visits must contain every pixel whose
    sampling coordinate passes the exact test (ix > -1 and ix < P, fp32, the expression of paste_axis_coord) - everything
    outside is only zero-filled.  Restated in numpy float32 and checked on random and adversarial box sides."""
    f = np.float32
    rng = np.random.default_rng(5)

    def exact_mask(b0, b1, P, n):
        pix = np.arange(n, dtype=np.float32)
        with np.errstate(all="ignore"):
            g = (pix + f(0.5) - f(b0)) / (f(b1) - f(b0)) * f(2) - f(1)
            ix = ((g + f(1)) * f(P) - f(1)) / f(2)
        return (ix > -1) & (ix < P)

    def reach(b0, b1, P, n):
        b0, b1 = f(b0), f(b1)
        with np.errstate(all="ignore"):
            bw = np.abs(b1 - b0)
            flo = np.fmin(b0, b1) - f(0.5) * bw / f(P) - f(2.5)
            fhi = np.fmax(b0, b1) + f(0.5) * bw / f(P) + f(1.5)
            lo = int(np.fmin(np.fmax(np.floor(flo), f(0)), f(n)))
            hi = int(np.fmax(np.fmin(np.ceil(fhi), f(n - 1)), f(-1)))
        return lo, hi

    cases = []
    for _ in range(3000):
        n = int(rng.integers(1, 1400))
        a = rng.uniform(-200, n + 200)
        w = rng.choice([rng.uniform(0, 2), rng.uniform(0, 60), rng.uniform(0, 2 * n + 1)])
        cases.append((a, a + w, int(rng.choice([7, 14, 28])), n))
    cases += [(0, 1333, 28, 1333), (-1e6, 1e6, 28, 800), (5.0, 5.0, 28, 100), (50.0, 20.0, 28, 100), (1e-3, 2e-3, 28, 64),
              (99.99, 100.0, 28, 100), (-30, -10, 28, 50), (60, 90, 28, 50), (float("inf"), 3.0, 28, 40), (-float("inf"), float("inf"), 28, 40),
              (float("nan"), 3.0, 28, 40), (0.0, 2.0 ** 24, 28, 1333), (-0.4999, 0.4999, 28, 9)]
    for b0, b1, P, n in cases:
        m = exact_mask(b0, b1, P, n)
        lo, hi = reach(b0, b1, P, n)
        idx = np.nonzero(m)[0]
        if idx.size:
            assert lo <= idx[0] and idx[-1] <= hi, (b0, b1, P, n, lo, hi, idx[0], idx[-1])


def test_select_foreground_proposals_stacked_equals_per_image_on_cpu():
    """The stacked (batched-gather) form of select_foreground_proposals is plain torch: the same check as the GPU test of
    tests/test_gpu_bookkeeping.py on the CPU - tables equal to per-image indexing of every column (roi_heads.py:46-75), incl.
    the sampler's lazy mask index and a column without a stacked form."""
    from u2seg_amd.modeling.batched import BatchList
    from u2seg_amd.modeling.roi_heads import select_foreground_proposals
    from u2seg_amd.structures import BitMasks, Boxes, Instances

    g = torch.Generator().manual_seed(4)
    nb, s, k, ng = 4, 48, 7, 6
    boxes, gtb = torch.rand((nb, s, 4), generator=g) * 100, torch.rand((nb, s, 4), generator=g) * 100
    cls = torch.randint(0, k + 1, (nb, s), generator=g)
    cls[0] = k
    cls[3, :3] = -1
    logits = torch.randn((nb, s), generator=g)
    match = torch.randint(0, ng, (nb, s), generator=g)
    bases = [torch.rand((ng, 9, 11), generator=g) > 0.5 for _ in range(nb)]
    extra = torch.randn((nb, s, 2), generator=g)

    def build(stacked):
        out = BatchList()
        for i in range(nb):
            r = Instances((9, 11))
            r.proposal_boxes, r.objectness_logits, r.gt_classes = Boxes(boxes[i]), logits[i], cls[i]
            r.gt_boxes, r.gt_masks, r.gt_extra = Boxes(gtb[i]), BitMasks(bases[i])[match[i]], extra[i]
            out.append(r)
        if stacked:
            out.boxes, out.gt_classes, out.gt_boxes, out.logits, out.match = boxes, cls, gtb, logits, match
        return out

    fg_s, m_s = select_foreground_proposals(build(True), k)
    fg_l, m_l = select_foreground_proposals(build(False), k)
    for a, b, ma, mb in zip(fg_s, fg_l, m_s, m_l):
        assert torch.equal(ma, mb) and len(a) == len(b) and sorted(a.get_fields()) == sorted(b.get_fields())
        for name in ("objectness_logits", "gt_classes", "gt_extra"):
            assert torch.equal(a.get(name), b.get(name)), name
        assert torch.equal(a.proposal_boxes.tensor, b.proposal_boxes.tensor) and torch.equal(a.gt_boxes.tensor, b.gt_boxes.tensor)
        assert torch.equal(a.gt_masks.tensor, b.gt_masks.tensor)
    assert len(fg_s[0]) == 0 and len(fg_s[1]) > 0


def test_host_thread_cap_and_image_index(monkeypatch):
    """utils/env.configure_host_threads caps torch's intra-op pool (never raises it, leaves an explicit OMP_NUM_THREADS alone) and
    reads the container's CPU quota; batched.image_index (numpy instead of ATen's CPU repeat_interleave, whose grain-1 parallel
    region wakes the whole OpenMP pool) still numbers the ROIs image by image."""
    import torch

    from u2seg_amd.modeling.batched import image_index
    from u2seg_amd.utils import env

    before = torch.get_num_threads()
    try:
        monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
        torch.set_num_threads(max(before, 2))
        monkeypatch.setattr(env, "cgroup_cpu_quota", lambda: 2.0)
        assert env.configure_host_threads(max_threads=8) == 1          # half of a 2-CPU quota
        torch.set_num_threads(1)
        monkeypatch.setattr(env, "cgroup_cpu_quota", lambda: None)
        assert env.configure_host_threads(max_threads=8) == 1          # never raised
        torch.set_num_threads(max(before, 2))
        monkeypatch.setenv("OMP_NUM_THREADS", "5")
        assert env.configure_host_threads(max_threads=1) == max(before, 2)   # an explicit setting is the user's
    finally:
        torch.set_num_threads(before)
    q = env.cgroup_cpu_quota()
    assert q is None or q > 0
    sizes = [3, 0, 2, 5]
    got = image_index(sizes, "cpu")
    want = torch.repeat_interleave(torch.arange(4, dtype=torch.float32), torch.tensor(sizes))
    assert got.dtype == torch.float32 and torch.equal(got, want)
