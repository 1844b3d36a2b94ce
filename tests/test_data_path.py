"""The real input pipeline against the reference: RLE codec on the strings the reference's
tests carry, transforms / annotation handling on the known answers of tests/data/test_detection_utils.py, and
registration -> dataset dicts -> DatasetMapper -> samplers against outputs of the reference itself on the committed small
dataset (tests/golden/data_small, fixture generator: tests/golden/make_fixtures.py --only data).  CPU only."""
import copy
import itertools
import json
import os

import numpy as np
import pytest
import torch

from u2seg_amd.config import get_cfg
from u2seg_amd.data import (DatasetCatalog, DatasetMapper, InferenceSampler, MetadataCatalog, TrainingSampler,
                            build_detection_test_loader, build_detection_train_loader, get_detection_dataset_dicts,
                            register_all_coco, rle)
from u2seg_amd.data import detection_utils as utils
from u2seg_amd.data import transforms as T
from u2seg_amd.data.build import AspectRatioGroupedDataset, _SampledStream, worker_init_reset_seed
from u2seg_amd.data.detection_utils import BoxMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DATA_ROOT = os.path.join(GOLD, "data_small")
CFG = os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml")


def donut():
    yy, xx = np.mgrid[0:100, 0:100]
    d = np.sqrt((xx - 50.0) ** 2 + (yy - 50.0) ** 2)
    return ((d > 10) & (d < 20)).astype(np.uint8)


def uncompressed_rle(mask):
    flat = mask.flatten(order="F").tolist()
    counts, prev, cnt = [], 0, 0
    for v in flat:
        if v == prev:
            cnt += 1
        else:
            counts.append(cnt)
            prev, cnt = v, 1
    counts.append(cnt)
    return {"counts": counts, "size": [mask.shape[0], mask.shape[1]]}


def test_rle_reference_vectors_and_round_trips():
    vecs = json.load(open(os.path.join(GOLD, "rle_vectors.json")))["vectors"]
    assert len(vecs) == 4
    for v in vecs:
        m = rle.decode(v)  # raises unless the decoded runs cover exactly h * w pixels
        assert m.shape == tuple(v["size"]) and m.dtype == np.uint8 and set(np.unique(m)) <= {0, 1}
        assert rle.encode(m)["counts"] == v["counts"]  # the exact string comes back
        assert rle.area(v) == int(m.sum())
        if "detection_bbox_xywh" in v:  # the mask of a detection lies inside its (slightly looser) predicted box
            x, y, w, h = rle.to_bbox(v)
            bx, by, bw, bh = v["detection_bbox_xywh"]
            assert bx - 2 <= x and by - 2 <= y and x + w <= bx + bw + 2 and y + h <= by + bh + 2
    # tests/data/test_coco.py: donut mask -> RLE -> mask, and uncompressed counts compress to the same string
    mask = donut()
    enc = rle.encode(mask)
    assert np.array_equal(rle.decode(enc), mask)
    assert rle.compress(uncompressed_rle(mask)) == enc
    assert np.array_equal(rle.decode(uncompressed_rle(mask)), mask)
    assert np.array_equal(rle.decode({"size": [3, 2], "counts": rle.encode(np.ones((3, 2)))["counts"]}), np.ones((3, 2)))
    assert rle.decode(rle.encode(np.zeros((5, 7), dtype=np.uint8))).sum() == 0
    with pytest.raises(ValueError):
        rle.decode({"size": [4, 4], "counts": [3, 2]})


def test_transform_known_answers():
    """tests/data/test_detection_utils.py:16-35, 89-134 and the box / size rules of augmentation_impl.py:180-201."""
    tfms = T.TransformList([T.HFlipTransform(400)])
    anno = {"bbox": np.asarray([10, 10, 200, 300]), "bbox_mode": BoxMode.XYXY_ABS, "category_id": 3,
            "segmentation": [[10, 10, 100, 100, 100, 10], [150, 150, 200, 150, 200, 200]]}
    out = utils.transform_instance_annotations(anno, tfms, (400, 400))
    assert np.allclose(out["bbox"], [200, 10, 390, 300])
    assert len(out["segmentation"]) == 2 and np.allclose(out["segmentation"][0], [390, 10, 300, 100, 300, 10])
    assert len(utils.annotations_to_instances([], (400, 400))) == 0
    # RLE mask through a flip (and a flip + resize): the left half becomes the right half
    mask = np.zeros((300, 400), dtype=np.uint8)
    mask[:, :200] = 1
    anno = {"bbox": np.asarray([10, 10, 200, 300]), "bbox_mode": BoxMode.XYXY_ABS, "segmentation": rle.encode(mask),
            "category_id": 3}
    out = utils.transform_instance_annotations(copy.deepcopy(anno), tfms, (300, 400))
    assert (out["segmentation"][:, 200:] == 1).all() and (out["segmentation"][:, :200] == 0).all()
    inst = utils.annotations_to_instances([out, out], (300, 400), mask_format="bitmask")
    assert inst.gt_masks.tensor.shape == (2, 300, 400) and inst.gt_classes.tolist() == [3, 3]
    both = T.TransformList([T.HFlipTransform(400), T.ResizeTransform(300, 400, 400, 400)])
    out = utils.transform_instance_annotations(copy.deepcopy(anno), both, (400, 400))
    assert out["segmentation"].shape == (400, 400) and np.allclose(out["bbox"], [200, 10 * 4 / 3, 390, 400])
    # no-ops are dropped, nested lists are flattened
    assert len(T.TransformList([T.NoOpTransform(), T.TransformList([T.HFlipTransform(5), T.NoOpTransform()])])) == 1
    # short edge to 80 capped at 133 (tests/data/test_transforms.py:245-258 uses these numbers)
    assert T.ResizeShortestEdge.get_output_shape(10, 10, 80, 133) == (80, 80)
    assert T.ResizeShortestEdge.get_output_shape(8, 100, 80, 133) == (11, 133)
    assert T.ResizeShortestEdge.get_output_shape(480, 640, 800, 1333) == (800, 1067)
    assert T.ResizeShortestEdge.get_output_shape(427, 640, 800, 1333) == (800, 1199)
    assert T.ResizeShortestEdge.get_output_shape(333, 1000, 800, 1333) == (444, 1333)
    # xywh -> xyxy of json floats happens in fp32, of ints stays int, arrays keep float64 (structures/boxes.py:62-131)
    assert BoxMode.convert([9.73, 19.42, 23.58, 28.57], BoxMode.XYWH_ABS, BoxMode.XYXY_ABS) == \
        [float(np.float32(9.73)), float(np.float32(19.42)), float(np.float32(9.73) + np.float32(23.58)),
         float(np.float32(19.42) + np.float32(28.57))]
    assert BoxMode.convert((1, 2, 3, 4), BoxMode.XYWH_ABS, BoxMode.XYXY_ABS) == (1, 2, 4, 6)
    assert BoxMode.convert(np.array([[1.5, 2, 3, 4]]), BoxMode.XYXY_ABS, BoxMode.XYWH_ABS).tolist() == [[1.5, 2, 1.5, 2]]
    with pytest.raises(utils.SizeMismatchError):
        utils.check_image_size({"width": 3, "height": 2, "file_name": "x"}, np.zeros((3, 2, 3)))


@pytest.fixture(scope="module")
def small(monkeypatch_module=None):
    """The builtin registration pointed at the committed small dataset (CLUSTER_NUM = 800), the reference-run listing and
    arrays, and the config with the fixture's INPUT overrides."""
    os.environ["CLUSTER_NUM"] = "800"
    for name in list(DatasetCatalog.keys()):
        DatasetCatalog.remove(name)
    for name in list(MetadataCatalog.keys()):
        MetadataCatalog.remove(name)
    register_all_coco(DATA_ROOT)
    listing = json.load(open(os.path.join(GOLD, "data_golden.json")))
    arrays = np.load(os.path.join(GOLD, "data_golden.npz"))
    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(listing["input_opts"] + ["DATALOADER.NUM_WORKERS", 0])
    return cfg, listing, arrays


def test_registration_and_dataset_dicts(small):
    cfg, fx, _ = small
    assert cfg.DATASETS.TRAIN[0] == fx["train_name"] and cfg.DATASETS.TEST[0] == fx["test_name"]
    assert fx["train_name"] in DatasetCatalog and fx["train_name"].replace("_separated", "_stuffonly") in DatasetCatalog
    assert len(DatasetCatalog.get(fx["train_name"])) == fx["num_raw"]
    dicts = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    assert len(dicts) == len(fx["dicts"])
    for got, want in zip(dicts, fx["dicts"]):
        got = copy.deepcopy(got)
        for k in ("file_name", "sem_seg_file_name"):
            got[k] = os.path.relpath(os.path.realpath(got[k]), os.path.realpath(DATA_ROOT))
        for a in got["annotations"]:
            a["bbox_mode"] = int(a["bbox_mode"])
        assert got == want
    # FILTER_EMPTY_ANNOTATIONS is off in the U2Seg configs; switched on, the crowd-only image goes
    kept = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=True)
    assert [d["image_id"] for d in kept] == [10, 20, 40]
    meta = MetadataCatalog.get(fx["train_name"])
    m = fx["meta"]
    assert meta.evaluator_type == m["evaluator_type"] and meta.ignore_label == m["ignore_label"]
    assert len(meta.thing_classes) == m["num_thing_classes"] and len(meta.stuff_classes) == m["num_stuff_classes"]
    assert meta.thing_dataset_id_to_contiguous_id[800] == m["thing_id_800"]
    assert meta.stuff_dataset_id_to_contiguous_id[801] == m["stuff_id_801"]
    assert os.path.relpath(os.path.realpath(meta.sem_seg_root), os.path.realpath(DATA_ROOT)) == m["sem_seg_root"]
    with pytest.raises(KeyError):
        DatasetCatalog.get("no_such_dataset")
    with pytest.raises(AssertionError):
        meta.ignore_label = 0  # metadata values never change silently


def test_dataset_mapper_matches_reference(small):
    """Every output of the reference's DatasetMapper on the small dataset - 5 seeded augmentation draws per image in
    training mode (10 short-edge choices incl. the max-size clamp, flips) and the test-mode resize - bit for bit: image,
    label map, boxes (fp32), classes, bitmasks."""
    cfg, fx, arrays = small
    dicts = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    train, test = DatasetMapper(cfg, True), DatasetMapper(cfg, False)
    flips = 0
    for case in fx["cases"]:
        key = case["key"]
        if key.startswith("test_"):
            out = test(dicts[case["index"]])
            assert "instances" not in out and "annotations" not in out
        else:
            np.random.seed(case["np_seed"])
            out = train(dicts[case["index"]])
            inst = out["instances"]
            assert list(inst.image_size) == case["image_size"]
            assert sorted(inst.get_fields().keys()) == case["instance_fields"]
            assert inst.gt_boxes.tensor.dtype == torch.float32
            assert np.array_equal(inst.gt_boxes.tensor.numpy(), arrays[key + "_boxes"])
            assert np.array_equal(inst.gt_classes.numpy(), arrays[key + "_classes"])
            if inst.has("gt_masks"):
                assert inst.gt_masks.tensor.dtype == torch.bool
                assert np.array_equal(np.packbits(inst.gt_masks.tensor.numpy(), axis=-1), arrays[key + "_masks"])
            assert out["image_id"] == case["image_id"]
        assert sorted(out.keys()) == case["keys"]
        assert (out["height"], out["width"]) == (case["height"], case["width"])  # the ORIGINAL size, for post-processing
        assert out["image"].dtype == torch.uint8 and out["sem_seg"].dtype == torch.int64
        assert np.array_equal(out["image"].numpy(), arrays[key + "_image"])
        assert np.array_equal(out["sem_seg"].numpy().astype(np.uint8), arrays[key + "_sem_seg"])
    # the mapper must not touch the dataset dict it was given
    before = copy.deepcopy(dicts[0])
    train(dicts[0])
    assert dicts[0] == before


def test_samplers_match_reference(small):
    _, fx, _ = small
    assert list(itertools.islice(iter(TrainingSampler(7, seed=11)), 30)) == fx["training_sampler_seed11_size7"]
    assert [list(InferenceSampler._get_local_indices(10, 3, r)) for r in range(3)] == fx["inference_shards_10_3"]
    # tests/data/test_sampler.py:98-111
    expect = {(0, 5): [range(0)] * 5, (16, 2): [range(8), range(8, 16)], (2, 3): [range(1), range(1, 2), range(0)],
              (42, 4): [range(11), range(11, 22), range(22, 32), range(32, 42)]}
    for (size, world), want in expect.items():
        assert [InferenceSampler._get_local_indices(size, world, r) for r in range(world)] == want
    # tests/data/test_sampler.py:38-74: a permutation per epoch; the unseeded sampler takes its seed from numpy's stream
    assert set(itertools.islice(iter(TrainingSampler(100, seed=10)), 100)) == set(range(100))
    np.random.seed(42)
    a = list(itertools.islice(iter(TrainingSampler(30)), 65))
    np.random.seed(42)
    s = TrainingSampler(30)
    np.random.seed(999)
    assert list(itertools.islice(iter(s), 65)) == a
    with pytest.raises(ValueError):
        TrainingSampler(0)
    # several loader workers neither duplicate nor reorder the stream (test_sampler.py:44-61)
    sampler = TrainingSampler(100, seed=10)
    want = list(itertools.islice(iter(sampler), 100))
    for workers in (0, 2):
        loader = torch.utils.data.DataLoader(_SampledStream(list(range(100)), sampler), num_workers=workers, batch_size=1,
                                             collate_fn=lambda b: b[0], worker_init_fn=worker_init_reset_seed)
        assert list(itertools.islice(iter(loader), 100)) == want
    # landscape / portrait buckets
    stream = [{"width": w, "height": h, "i": i} for i, (w, h) in enumerate([(4, 3), (3, 4), (5, 3), (3, 3), (2, 5), (9, 1)])]
    batches = list(AspectRatioGroupedDataset(stream, 2))
    assert [[d["i"] for d in b] for b in batches] == [[0, 2], [1, 3]]


def test_train_and_test_loaders(small):
    cfg, fx, _ = small
    cfg = cfg.clone()
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 2])
    loader = build_detection_train_loader(cfg, seed=5)
    batches = list(itertools.islice(iter(loader), 4))
    for b in batches:
        assert len(b) == 2
        assert len({d["width"] > d["height"] for d in b}) == 1  # one aspect-ratio group per batch
        for d in b:
            assert d["image"].shape[1:] == tuple(d["instances"].image_size) == tuple(d["sem_seg"].shape)
    # same seed, same numpy stream -> the same batches again
    np.random.seed(3)
    first = [d["image_id"] for b in itertools.islice(iter(build_detection_train_loader(cfg, seed=5)), 3) for d in b]
    np.random.seed(3)
    again = [d["image_id"] for b in itertools.islice(iter(build_detection_train_loader(cfg, seed=5)), 3) for d in b]
    assert first == again
    test_loader = build_detection_test_loader(cfg, cfg.DATASETS.TRAIN[0])
    seen = [b[0]["image_id"] for b in test_loader]
    assert seen == [d["image_id"] for d in fx["dicts"]]


def test_prefetcher_staging_logic(small):
    """DevicePrefetcher's packing into reusable staging blocks (here unpinned, target "cpu"; the GPU test runs the pinned /
    side-stream form): five batches through three blocks come out equal to the loader's own output, label maps keep int64."""
    from u2seg_amd.data import DevicePrefetcher

    cfg, _, _ = small
    cfg = cfg.clone()
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 2])
    np.random.seed(1)
    host = list(itertools.islice(iter(build_detection_train_loader(cfg, seed=3)), 5))
    np.random.seed(1)
    staged = list(itertools.islice(iter(DevicePrefetcher(build_detection_train_loader(cfg, seed=3), "cpu", _pinned=False)), 5))
    assert len(staged) == 5
    for hb, sb in zip(host, staged):
        for h, d in zip(hb, sb):
            assert sorted(h.keys()) == sorted(d.keys()) and h["file_name"] == d["file_name"]
            assert torch.equal(h["image"], d["image"]) and torch.equal(h["sem_seg"], d["sem_seg"])
            assert d["sem_seg"].dtype == torch.int64
            hi, di = h["instances"], d["instances"]
            assert sorted(hi.get_fields()) == sorted(di.get_fields()) and hi.image_size == di.image_size
            assert torch.equal(hi.gt_boxes.tensor, di.gt_boxes.tensor) and torch.equal(hi.gt_classes, di.gt_classes)
            if hi.has("gt_masks"):
                assert torch.equal(hi.gt_masks.tensor, di.gt_masks.tensor)
    with pytest.raises(AssertionError):
        DevicePrefetcher([], "cpu")  # the product form needs a GPU


def test_slot_transport_round_trip(small):
    """data/slots.py: a mapped batch packed into a shared-memory slot (label maps as bytes, masks as bits) and rebuilt
    from the block equals the original, field for field; an oversized batch falls back to the plain list; with loader
    workers the batches keep one orientation each and arrive complete."""
    from u2seg_amd.data.build import BatchIndexStream
    from u2seg_amd.data.slots import BatchPacker, PackedBatch, SlotRing, unpack

    cfg, _, _ = small
    cfg = cfg.clone()
    cfg.merge_from_list(["SOLVER.IMS_PER_BATCH", 4, "DATALOADER.ASPECT_RATIO_GROUPING", False])
    np.random.seed(7)
    batches = list(itertools.islice(iter(build_detection_train_loader(cfg, seed=2)), 3))
    ring = SlotRing(1, 4 << 20, 3)
    packer = BatchPacker(ring)
    for k, batch in enumerate(batches):
        packed = packer(batch)
        assert isinstance(packed, PackedBatch) and packed.slot == k % 3 and packed.nbytes % 256 == 0
        rebuilt = unpack(torch.from_numpy(ring.view(packed.slot)[: packed.nbytes].copy()), packed.samples)
        for h, d in zip(batch, rebuilt):
            assert list(h.keys()) == list(d.keys())
            for key in ("file_name", "height", "width", "image_id"):
                assert h[key] == d[key]
            assert torch.equal(h["image"], d["image"]) and d["image"].dtype == torch.uint8
            assert torch.equal(h["sem_seg"], d["sem_seg"]) and d["sem_seg"].dtype == torch.int64
            hi, di = h["instances"], d["instances"]
            assert list(hi.get_fields()) == list(di.get_fields()) and hi.image_size == di.image_size
            assert torch.equal(hi.gt_boxes.tensor, di.gt_boxes.tensor) and di.gt_boxes.tensor.dtype == torch.float32
            assert torch.equal(hi.gt_classes, di.gt_classes) and di.gt_classes.dtype == torch.int64
            if hi.has("gt_masks"):
                assert di.gt_masks.tensor.dtype == torch.bool and torch.equal(hi.gt_masks.tensor, di.gt_masks.tensor)
    assert BatchPacker(SlotRing(1, 1024, 3))(batches[0]) is batches[0]  # does not fit: plain list
    assert BatchPacker(None)(batches[0]) is batches[0]
    # index-level grouping == AspectRatioGroupedDataset on the mapped stream
    land = [True, False, True, False, False, True]
    assert list(BatchIndexStream([0, 1, 2, 3, 4, 5], 2, land)) == [[0, 2], [1, 3]]
    assert list(BatchIndexStream(range(5), 2)) == [[0, 1], [2, 3]]
    # through worker processes
    cfg.merge_from_list(["DATALOADER.NUM_WORKERS", 2, "DATALOADER.ASPECT_RATIO_GROUPING", True, "SOLVER.IMS_PER_BATCH", 2])
    loader = build_detection_train_loader(cfg, seed=5)
    got = list(itertools.islice(iter(loader), 6))
    sampler_order = list(itertools.islice(iter(TrainingSampler(4, seed=5)), 40))
    ids = [10, 20, 30, 40]  # dataset order; 20 is the portrait image, 30 the square one (counts as portrait: w > h fails)
    want = list(itertools.islice(iter(BatchIndexStream(sampler_order, 2, [True, False, False, True])), 6))
    assert [[d["image_id"] for d in b] for b in got] == [[ids[i] for i in b] for b in want]
    for b in got:
        assert len({d["width"] > d["height"] for d in b}) == 1
        for d in b:
            assert d["image"].shape[1:] == tuple(d["instances"].image_size) == tuple(d["sem_seg"].shape)


def test_pseudo_panoptic_merge_matches_reference_script(tmp_path):
    """u2seg_amd.data.pseudo_panoptic.generate == the reference script generate_pseudo_panoptic.py run on the same tree
    (tests/golden/make_fixtures.py --only pseudo_panoptic): the json (segment ids running on across images, the covered
    instance dropped, the mostly-hidden semantic class skipped, the image without pseudo instances left out) and every
    pixel of the id maps."""
    from PIL import Image

    from u2seg_amd.data import pseudo_panoptic as PP

    fx = json.load(open(os.path.join(GOLD, "pseudo_panoptic_golden.json")))
    arrays = np.load(os.path.join(GOLD, "pseudo_panoptic_golden.npz"))
    ann_root = tmp_path / "datasets" / "prepare_ours" / "u2seg_annotations"
    sem_dir = ann_root / "semantic_annotations" / "stego_coco_train_semantic_seg_resized"
    for d in (ann_root / "ins_annotations", sem_dir, ann_root / "panoptic_annotations",
              tmp_path / "datasets" / "datasets" / "panoptic_anns"):
        os.makedirs(d)
    json.dump(fx["template"], open(tmp_path / "datasets" / "datasets" / "panoptic_anns" / "panoptic_train2017.json", "w"))
    json.dump(fx["pseudo"], open(ann_root / "ins_annotations" / "cocotrain_800_ins_panoptic.json", "w"))
    with open(ann_root / "semantic_annotations" / "coco_train_img_file_names.txt", "w") as f:
        for i, name in enumerate(fx["names"]):
            f.write(name + "\n")
            np.save(sem_dir / ("%d.npy" % i), arrays["semantic_%d" % i])
    out = PP.generate(str(tmp_path), 800, "train")
    assert out == fx["expected"]
    assert json.load(open(ann_root / "panoptic_annotations" / "cocotrain_800.json")) == fx["expected"]
    assert [a["image_id"] for a in out["annotations"]] == [100, 101] and [im["id"] for im in out["images"]] == [100, 101]
    for a in out["annotations"]:
        png = np.asarray(Image.open(ann_root / "panoptic_annotations" / "cocotrain_800" / a["file_name"]))
        ids = PP.rgb2id(png)
        assert np.array_equal(ids, arrays["ids_" + a["file_name"]])
        assert set(np.unique(ids).tolist()) - {0} == {s["id"] for s in a["segments_info"]}
    assert len(out["categories"]) == 827 and out["categories"][799]["isthing"] == 1 and out["categories"][800]["isthing"] == 0
    big = np.array([[0, 255, 256, 70000]], dtype=np.uint32)
    assert np.array_equal(PP.rgb2id(PP.id2rgb(big)), big)


def test_label_preparation_steps_match_reference_scripts(tmp_path):
    """The remaining datasets/prepare_ours steps against the reference's own code (fixture: make_fixtures.py --only
    label_prep): panoptic png -> semantic label maps (chained onto the pseudo-panoptic golden), cluster ids into the
    class-agnostic instance annotations, supercategory ids into the ground-truth panoptic json."""
    import copy as _copy

    from PIL import Image

    from u2seg_amd.data import pseudo_panoptic as PP

    fx = json.load(open(os.path.join(GOLD, "label_prep_golden.json")))
    sem = np.load(os.path.join(GOLD, "label_prep_golden.npz"))
    pan = json.load(open(os.path.join(GOLD, "pseudo_panoptic_golden.json")))["expected"]
    ids = np.load(os.path.join(GOLD, "pseudo_panoptic_golden.npz"))
    pan_root, sem_root = tmp_path / "pan", tmp_path / "sem"
    os.makedirs(pan_root)
    json.dump(pan, open(tmp_path / "pan.json", "w"))
    for a in pan["annotations"]:
        Image.fromarray(PP.id2rgb(ids["ids_" + a["file_name"]])).save(pan_root / a["file_name"])
    assert PP.separate_semantic_from_panoptic(str(tmp_path / "pan.json"), str(pan_root), str(sem_root), pan["categories"]) == 2
    for a in pan["annotations"]:
        got = np.asarray(Image.open(sem_root / a["file_name"]))
        assert got.dtype == np.uint8 and np.array_equal(got, sem["sem_" + a["file_name"]])
    c = fx["classaware"]
    assert PP.classaware_instance_annotations(_copy.deepcopy(c["template"]), c["clusters"], _copy.deepcopy(c["masks"])) == c["expected"]
    assert [im["id"] for im in c["expected"]["images"]] == [7, 9] and len(c["expected"]["categories"]) == 300
    s = fx["supercategory"]
    for n in (300, 800):
        assert PP.panoptic_supercategory_annotations(_copy.deepcopy(s["standard"]), n) == s["expected"][str(n)]
    from u2seg_amd.evaluation.sem_seg_evaluation import STUFF_TO_SUPERCATEGORY

    assert tuple(PP.STUFF_ID_TO_SUPERCATEGORY.values()) == STUFF_TO_SUPERCATEGORY  # one table, two orderings


def test_rle_and_transform_properties():
    """Size-independent properties (hypothesis): decode(encode(m)) == m and compress(uncompressed(m)) == encode(m) for random
    masks incl. empty / full / single-column ones; area and tight box agree with the mask; flipping twice is the identity on
    images, boxes and masks; a resize maps the image corners onto the new corners."""
    from hypothesis import given, settings
    from hypothesis import strategies as st
    from hypothesis.extra import numpy as hnp

    @settings(max_examples=60, deadline=None)
    @given(hnp.arrays(np.uint8, st.tuples(st.integers(1, 17), st.integers(1, 23)), elements=st.integers(0, 1)))
    def codec(mask):
        enc = rle.encode(mask)
        assert np.array_equal(rle.decode(enc), mask) and enc["size"] == list(mask.shape)
        assert rle.compress(uncompressed_rle(mask)) == enc
        assert rle.area(enc) == int(mask.sum())
        x, y, w, h = rle.to_bbox(enc)
        if mask.any():
            ys, xs = np.nonzero(mask)
            assert (x, y, w, h) == (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1)
        else:
            assert (x, y, w, h) == (0, 0, 0, 0)

    codec()

    @settings(max_examples=30, deadline=None)
    @given(st.integers(2, 40), st.integers(2, 40), st.integers(0, 2 ** 31 - 1))
    def geometry(h, w, seed):
        rs = np.random.RandomState(seed)
        img = rs.randint(0, 255, (h, w, 3)).astype(np.uint8)
        box = np.array([[rs.uniform(0, w / 2), rs.uniform(0, h / 2), rs.uniform(w / 2, w), rs.uniform(h / 2, h)]])
        twice = T.TransformList([T.HFlipTransform(w), T.HFlipTransform(w)])
        assert np.array_equal(twice.apply_image(img), img) and np.allclose(twice.apply_box(box.copy()), box)
        one = T.HFlipTransform(w).apply_box(box.copy())
        assert np.allclose(one[:, [0, 2]], w - box[:, [2, 0]]) and np.allclose(one[:, [1, 3]], box[:, [1, 3]])
        nh, nw = T.ResizeShortestEdge.get_output_shape(h, w, 32, 50)
        assert max(nh, nw) <= 50 and (min(nh, nw) == 32 or max(nh, nw) == 50)
        rt = T.ResizeTransform(h, w, nh, nw)
        assert rt.apply_image(img).shape == (nh, nw, 3)
        assert np.allclose(rt.apply_box(np.array([[0.0, 0.0, w, h]])), [[0, 0, nw, nh]])

    geometry()


def test_slot_packing_of_arbitrary_batches():
    """BatchPacker / unpack on batches that do not come from the mapper: zero instances with an empty bool mask tensor,
    odd widths (bit packing pads every row to whole bytes), label maps that do not fit a byte (stay int64), tiny label maps,
    float images, extra python values - everything comes back with its dtype, shape and values."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from u2seg_amd.data.slots import BatchPacker, PackedBatch, SlotRing, unpack
    from u2seg_amd.structures import BitMasks, Boxes, Instances

    ring = SlotRing(1, 8 << 20, 2)

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 40), st.integers(1, 40), st.integers(0, 5), st.integers(0, 2 ** 31 - 1), st.booleans())
    def round_trip(h, w, n, seed, wide_labels):
        g = torch.Generator().manual_seed(seed)
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(torch.rand((n, 4), generator=g) * 30)
        inst.gt_classes = torch.randint(0, 800, (n,), generator=g)
        inst.gt_masks = BitMasks(torch.rand((n, h, w), generator=g) > 0.5)
        sem = torch.randint(0, 70000 if wide_labels else 256, (h * 3, w * 37), generator=g)  # >= 4096 elements in most draws
        sample = {"file_name": "a/b.jpg", "height": h, "width": w, "image_id": seed,
                  "image": torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8),
                  "aux": torch.rand((2, 3), generator=g), "sem_seg": sem, "instances": inst, "extra": {"k": [1, 2]}}
        packer = BatchPacker(ring)
        packed = packer([sample, sample])
        assert isinstance(packed, PackedBatch)
        for d in unpack(torch.from_numpy(ring.view(packed.slot)[: packed.nbytes].copy()), packed.samples):
            assert list(d.keys()) == list(sample.keys()) and d["extra"] == {"k": [1, 2]} and d["file_name"] == "a/b.jpg"
            for key in ("image", "aux", "sem_seg"):
                assert d[key].dtype == sample[key].dtype and torch.equal(d[key], sample[key]), key
            di = d["instances"]
            assert di.image_size == (h, w) and len(di) == n
            assert torch.equal(di.gt_boxes.tensor, inst.gt_boxes.tensor) and torch.equal(di.gt_classes, inst.gt_classes)
            assert di.gt_masks.tensor.dtype == torch.bool and di.gt_masks.tensor.shape == (n, h, w)
            assert torch.equal(di.gt_masks.tensor, inst.gt_masks.tensor)

    round_trip()
