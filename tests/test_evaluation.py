"""U2Seg evaluation front-end against the reference's own evaluators run on a tiny validation
set (tests/golden/make_fixtures.py --only eval): instance cluster -> category mapping, semantic votes -> mapping ->
remapped confusion matrix -> metrics.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from u2seg_amd.data import DatasetCatalog, MetadataCatalog, register_coco_instances
from u2seg_amd.data.datasets import load_sem_seg
from u2seg_amd.evaluation import COCOEvaluator, DatasetEvaluators, SemSegEvaluator, inference_on_dataset, instances_to_coco_json
from u2seg_amd.evaluation import hungarian
from u2seg_amd.evaluation.sem_seg_evaluation import to_supercategories
from u2seg_amd.structures import Boxes, Instances

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture()
def tiny_val(tmp_path, monkeypatch):
    fx = json.load(open(os.path.join(GOLD, "eval_golden.json")))
    arrays = np.load(os.path.join(GOLD, "eval_golden.npz"))
    img_dir, gt_dir = tmp_path / "images", tmp_path / "sem_gt"
    os.makedirs(img_dir)
    os.makedirs(gt_dir)
    for im in fx["images"]:
        stem = im["file_name"][:-4]
        Image.fromarray(np.zeros((im["height"], im["width"], 3), dtype=np.uint8)).save(img_dir / im["file_name"])
        Image.fromarray(arrays["gt_" + stem], mode="L").save(gt_dir / (stem + ".png"))
    json_file = str(tmp_path / "val.json")
    json.dump({"images": fx["images"], "annotations": fx["annotations"], "categories": fx["categories"]}, open(json_file, "w"))
    for name in ("tiny_val", "tiny_val_sem"):
        if name in DatasetCatalog:
            DatasetCatalog.remove(name)
        if name in MetadataCatalog:
            MetadataCatalog.remove(name)
    register_coco_instances("tiny_val", {}, json_file, str(img_dir))
    DatasetCatalog.get("tiny_val")
    DatasetCatalog.register("tiny_val_sem", lambda: load_sem_seg(str(gt_dir), str(img_dir)))
    MetadataCatalog.get("tiny_val_sem").set(stuff_classes=[str(c) for c in range(28)], ignore_label=255)
    monkeypatch.chdir(tmp_path)  # the mapping files go to ./hungarian_matching like in the reference
    inputs, outputs = [], []
    for im, p in zip(fx["images"], fx["predictions"]):
        inst = Instances((im["height"], im["width"]))
        inst.pred_boxes = Boxes(torch.tensor(p["boxes"], dtype=torch.float32))
        inst.scores = torch.tensor(p["scores"], dtype=torch.float32)
        inst.pred_classes = torch.tensor(p["classes"], dtype=torch.int64)
        outputs.append({"instances": inst, "sem_seg": torch.from_numpy(arrays["logits_" + im["file_name"][:-4]])})
        inputs.append({"image_id": im["id"], "file_name": str(img_dir / im["file_name"]), "height": im["height"],
                       "width": im["width"]})
    return fx, inputs, outputs


def test_instance_cluster_mapping_matches_reference(tiny_val):
    fx, inputs, outputs = tiny_val
    ev = COCOEvaluator("tiny_val", mode="hungarian_matching")
    ev.process(inputs, outputs)
    results = [r for p in ev._predictions for r in p["instances"]]
    assert results == fx["coco_results"]  # xyxy -> xywh in the json, fp32 values as python floats
    out = ev.evaluate()
    assert {str(k): v for k, v in out["instance_mapping"].items()} == fx["instance_mapping"]
    assert json.load(open("./hungarian_matching/instance_mapping.json")) == fx["instance_mapping_file"]
    assert len(out["instance_mapping"]) == 300 and sum(v != -1 for v in out["instance_mapping"].values()) == 11
    # eval mode: clusters without a mapping disappear, the others carry the dataset id of their category
    ev2 = COCOEvaluator("tiny_val", output_dir="out", mode="eval")
    ev2.process(inputs, outputs)
    res = ev2.evaluate()["bbox"]
    mapping = hungarian.load_mapping("./hungarian_matching/instance_mapping.json")
    to_dataset = {v: k for k, v in MetadataCatalog.get("tiny_val").thing_dataset_id_to_contiguous_id.items()}
    want = [dict(r, category_id=to_dataset[mapping[r["category_id"]]]) for r in fx["coco_results"]
            if mapping[r["category_id"]] != -1]
    assert json.load(open("out/coco_instances_results.json")) == want
    assert res["num_results"] == len(want) and res["num_dropped"] == len(fx["coco_results"]) - len(want)


def test_semantic_mapping_and_metrics_match_reference(tiny_val):
    fx, inputs, outputs = tiny_val
    assert to_supercategories(np.concatenate([np.arange(54), [255]]).astype(int)).tolist() == fx["transfer_table"]
    ev = SemSegEvaluator("tiny_val_sem", mode="hungarian_matching")
    ev.process(inputs, outputs)
    assert sorted(zip(ev.pred_det_cate, ev.pseudo_gt_cate)) == [tuple(v) for v in fx["semantic_votes"]]
    out = ev.evaluate()
    assert out["sem_seg"] is None
    on_disk = json.load(open("./hungarian_matching/semantic_mapping.json"))
    assert on_disk == fx["semantic_mapping_file"] and list(on_disk) == list(fx["semantic_mapping_file"])  # same key order
    ev2 = SemSegEvaluator("tiny_val_sem", mode="eval")
    ev2.process(inputs, outputs)
    assert ev2._conf_matrix.tolist() == fx["conf_matrix"]  # incl. the reference's chained in-place remapping
    res = ev2.evaluate()["sem_seg"]
    assert list(res) == list(fx["sem_seg_results"])
    for k, v in fx["sem_seg_results"].items():
        if v is None:
            assert res[k] != res[k], k  # NaN for classes that do not occur
        else:
            assert res[k] == pytest.approx(v, rel=1e-12), k


@pytest.mark.gpu
def test_semantic_evaluator_accumulates_on_the_device(tiny_val):
    """evaluation/sem_seg_evaluation.py:205-300 with the predictions resident on the GPU (where inference leaves them): the
    joint (prediction, ground-truth) histogram of the voting pass and the 17 x 17 confusion matrix are accumulated on the
    device (torch.bincount there), only votes / the final matrix come back - same votes, mapping file, matrix and metrics as
    the reference-generated fixture."""
    fx, inputs, outputs = tiny_val
    assert torch.cuda.is_available()
    outputs = [dict(o, sem_seg=o["sem_seg"].to("cuda:0")) for o in outputs]
    ev = SemSegEvaluator("tiny_val_sem", mode="hungarian_matching")
    ev.process(inputs, outputs)
    assert sorted(zip(ev.pred_det_cate, ev.pseudo_gt_cate)) == [tuple(v) for v in fx["semantic_votes"]]
    assert ev.evaluate()["sem_seg"] is None
    assert json.load(open("./hungarian_matching/semantic_mapping.json")) == fx["semantic_mapping_file"]
    ev2 = SemSegEvaluator("tiny_val_sem", mode="eval")
    ev2.process(inputs, outputs)
    assert ev2._conf_matrix.is_cuda  # the accumulation stayed on the device
    assert ev2._conf_matrix.tolist() == fx["conf_matrix"]
    res = ev2.evaluate()["sem_seg"]
    for k, v in fx["sem_seg_results"].items():
        if v is None:
            assert res[k] != res[k], k
        else:
            assert res[k] == pytest.approx(v, rel=1e-12), k


def test_vote_mapping_and_loop():
    # coco_evaluation.py:273-297: majority per cluster, -1 without votes, ties to the smaller category
    m = hungarian.majority_vote_mapping([0, 0, 0, 2, 2, 5], [3, 3, 1, 4, 7, 0], range(4), 8)
    assert m == {0: 3, 1: -1, 2: 4, 3: -1}
    assert hungarian.majority_vote_mapping([], [], range(2), 3) == {0: -1, 1: -1}
    iou = hungarian.box_iou_xywh([0, 0, 10, 10], [[0, 0, 10, 10], [5, 5, 10, 10], [20, 20, 5, 5]])
    assert np.allclose(iou, [1.0, 25 / 175, 0.0])
    inst = Instances((4, 6))
    inst.pred_boxes = Boxes(torch.tensor([[1.0, 1.0, 4.0, 3.0]]))
    inst.scores = torch.tensor([0.5])
    inst.pred_classes = torch.tensor([7])
    inst.pred_masks = torch.zeros((1, 4, 6), dtype=torch.bool)
    inst.pred_masks[0, 1:3, 1:4] = True
    (r,) = instances_to_coco_json(inst, 9)
    assert r["bbox"] == [1.0, 1.0, 3.0, 2.0] and r["category_id"] == 7 and r["image_id"] == 9
    from u2seg_amd.data import rle

    assert rle.decode(r["segmentation"]).sum() == 6 and r["segmentation"]["size"] == [4, 6]

    class Model(torch.nn.Module):
        def forward(self, batch):
            return [{"v": x["v"] * 2} for x in batch]

    class Sum(DatasetEvaluators.__mro__[1]):
        def reset(self):
            self.total = 0

        def process(self, inputs, outputs):
            self.total += sum(o["v"] for o in outputs)

        def evaluate(self):
            return {"sum": self.total}

    model = Model().train()
    assert inference_on_dataset(model, [[{"v": 1}, {"v": 2}], [{"v": 3}]], Sum()) == {"sum": 12}
    assert model.training


def test_panoptic_conversion_matches_reference(tiny_val, tmp_path, monkeypatch):
    """COCOPanopticEvaluator.process in both modes == the reference's (fixture "panoptic_eval" / "panoptic_matching"): with
    the mapping files present, thing clusters become dataset category ids, stuff classes 300 + supercategory, segments of
    unmapped clusters are erased from the id map; without them the predictions pass through."""
    import io

    from u2seg_amd.data.pseudo_panoptic import rgb2id
    from u2seg_amd.evaluation import COCOPanopticEvaluator

    fx, inputs, _ = tiny_val
    os.makedirs("hungarian_matching")
    json.dump(fx["instance_mapping_file"], open("hungarian_matching/instance_mapping.json", "w"))
    json.dump(fx["semantic_mapping_file"], open("hungarian_matching/semantic_mapping.json", "w"))
    MetadataCatalog.get("tiny_val_pan").set(
        thing_dataset_id_to_contiguous_id=dict(MetadataCatalog.get("tiny_val").thing_dataset_id_to_contiguous_id))

    def outputs():
        return [{"panoptic_seg": (torch.tensor(p["ids"], dtype=torch.int32), [dict(s) for s in p["segments_info"]])}
                for p in fx["panoptic_inputs"]]

    def decoded(ev):
        return [{"image_id": p["image_id"], "file_name": p["file_name"], "segments_info": p["segments_info"],
                 "ids": rgb2id(np.asarray(Image.open(io.BytesIO(p["png_string"])))).tolist()} for p in ev._predictions]

    ev = COCOPanopticEvaluator("tiny_val_pan")
    assert ev.mode == "eval"
    ev.process(inputs, outputs())
    assert decoded(ev) == fx["panoptic_eval"]
    erased = [p for p, q in zip(fx["panoptic_eval"], fx["panoptic_inputs"]) if len(p["segments_info"]) < len(q["segments_info"])]
    assert erased and all(0 in np.unique(p["ids"]) for p in erased)  # cluster 298 has no mapping: its pixels are void
    res = ev.evaluate()["panoptic_seg"]
    saved = json.load(open(res["predictions_json"]))
    assert [a["file_name"] for a in saved["annotations"]] == [p["file_name"] for p in fx["panoptic_eval"]]
    assert res["num_images"] == len(fx["images"])
    monkeypatch.chdir(tmp_path / "images")  # a directory without mapping files
    ev2 = COCOPanopticEvaluator("tiny_val_pan")
    assert ev2.mode == "hungarian_matching"
    ev2.process(inputs, outputs())
    assert decoded(ev2) == fx["panoptic_matching"]


def test_two_pass_evaluation_through_train_net(tmp_path, monkeypatch):
    """tools/train_net.py's evaluation entry over the builtin validation registration: the tiny validation set laid out at
    the paths `coco_2017_val_panoptic_separated` expects under $DETECTRON2_DATASETS, a stub model replaying the fixture's
    predictions, first `hungarian_matching` then `eval` - the mapping files and the semantic metrics equal the
    reference-generated ones, and the three evaluators (semantic, instances, panoptic) are all assembled."""
    import importlib.util

    from u2seg_amd.config import get_cfg

    fx = json.load(open(os.path.join(GOLD, "eval_golden.json")))
    arrays = np.load(os.path.join(GOLD, "eval_golden.npz"))
    root = tmp_path / "data"
    img_dir = root / "coco" / "val2017"
    sem_dir = root / "datasets" / "panoptic_anns" / "panoptic_stuff_val2017"
    os.makedirs(img_dir)
    os.makedirs(sem_dir)
    os.makedirs(root / "coco" / "annotations")
    images = []
    for im in fx["images"]:
        stem = im["file_name"][:-4]
        Image.fromarray(np.zeros((im["height"], im["width"], 3), dtype=np.uint8)).save(img_dir / im["file_name"])
        Image.fromarray(arrays["gt_" + stem], mode="L").save(sem_dir / (stem + ".png"))
        images.append(im)
    # the builtin registration announces the 800 cluster categories "1".."800"; the json has to agree with it (metadata may
    # not change once set - the reference asserts the same), and ids 1..90 keep their contiguous ids 0..89
    cats = [{"id": c, "name": str(c), "supercategory": str(c)} for c in range(1, 801)]
    json.dump({"images": images, "annotations": fx["annotations"], "categories": cats},
              open(root / "coco" / "annotations" / "instances_val2017.json", "w"))
    monkeypatch.setenv("DETECTRON2_DATASETS", str(root))
    monkeypatch.setenv("CLUSTER_NUM", "800")
    monkeypatch.chdir(tmp_path)
    for cat in (DatasetCatalog, MetadataCatalog):
        for name in list(cat.keys()):
            cat.remove(name)
    spec = importlib.util.spec_from_file_location("u2seg_train_net", os.path.join(os.path.dirname(GOLD), "..", "tools", "train_net.py"))
    train_net = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(train_net)
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(os.path.dirname(GOLD), "..", "configs", "COCO-PanopticSegmentation", "u2seg_eval_800.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "DATALOADER.NUM_WORKERS", 0, "OUTPUT_DIR", str(tmp_path / "out")])
    by_id = {im["id"]: k for k, im in enumerate(fx["images"])}

    class Replay(torch.nn.Module):
        def forward(self, batch):
            outs = []
            for x in batch:
                k = by_id[x["image_id"]]
                im, p, pan = fx["images"][k], fx["predictions"][k], fx["panoptic_inputs"][k]
                inst = Instances((im["height"], im["width"]))
                inst.pred_boxes = Boxes(torch.tensor(p["boxes"], dtype=torch.float32))
                inst.scores = torch.tensor(p["scores"], dtype=torch.float32)
                inst.pred_classes = torch.tensor(p["classes"], dtype=torch.int64)
                outs.append({"instances": inst, "sem_seg": torch.from_numpy(arrays["logits_" + im["file_name"][:-4]]),
                             "panoptic_seg": (torch.tensor(pan["ids"], dtype=torch.int32), [dict(s) for s in pan["segments_info"]])})
            return outs

    model = Replay()
    name = cfg.DATASETS.TEST[0]
    first = train_net.evaluate_on_disk_datasets(cfg, model, "hungarian_matching", "cpu")[name]
    assert {str(k): v for k, v in first["instance_mapping"].items()} == fx["instance_mapping"]
    assert json.load(open("hungarian_matching/semantic_mapping.json")) == fx["semantic_mapping_file"]
    second = train_net.evaluate_on_disk_datasets(cfg, model, "eval", "cpu")[name]
    assert set(second) == {"sem_seg", "bbox", "panoptic_seg"}
    for k, v in fx["sem_seg_results"].items():
        if v is not None:
            assert second["sem_seg"][k] == pytest.approx(v, rel=1e-12), k
    assert second["panoptic_seg"]["num_images"] == len(images) and os.path.isfile(second["panoptic_seg"]["predictions_json"])
    assert second["bbox"]["num_results"] > 0
    assert all(k in second["bbox"] for k in ("AP", "AP50", "AP75", "APs", "APm", "APl")) and 0 <= second["bbox"]["AP50"] <= 100


def test_cocoeval_core_matches_reference_cpp():
    """evaluation/cocoeval.py's matching + accumulation == the reference's C++ COCOevalEvaluateImages / COCOevalAccumulate
    (fixture: make_fixtures.py --only cocoeval): every entry of the precision [10, 101, 5, 4, 3], recall [10, 5, 4, 3] and
    score tables, on a problem with crowd regions, all area classes, tied scores and a cell beyond the 100-detection budget.
    Plus hand-checkable cases for the IoU rule and the summary."""
    from u2seg_amd.evaluation import cocoeval as CE

    fx = json.load(open(os.path.join(GOLD, "cocoeval_golden.json")))
    ref = np.load(os.path.join(GOLD, "cocoeval_golden.npz"))
    out = CE.evaluate_bbox(fx["dataset"], fx["results"])
    assert out["precision"].shape == ref["precision"].shape == (10, 101, 5, 4, 3)
    np.testing.assert_allclose(out["precision"], ref["precision"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out["recall"], ref["recall"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out["scores"], ref["scores"], rtol=0, atol=1e-12)
    p = ref["precision"][:, :, :, 0, 2]
    assert out["stats"]["AP"] == pytest.approx(float(p[p > -1].mean()), rel=1e-12)
    # crowd ground truth: the union is the detection's own area (a detection inside a crowd region has IoU 1)
    iou = CE.box_ious([[10, 10, 10, 10]], [[0, 0, 100, 100], [0, 0, 100, 100]], [1, 0])
    assert np.allclose(iou, [[1.0, 100 / 10000]])
    # one image, two instances, detections: exact hit (0.9), shifted hit (0.8, IoU 0.905), stray (0.7)
    gt = {"images": [{"id": 1}], "categories": [{"id": 1}],
          "annotations": [{"id": 1, "image_id": 1, "category_id": 1, "bbox": [10, 10, 20, 20], "area": 400, "iscrowd": 0},
                          {"id": 2, "image_id": 1, "category_id": 1, "bbox": [50, 50, 40, 40], "area": 1600, "iscrowd": 0}]}
    res = [{"image_id": 1, "category_id": 1, "bbox": [10, 10, 20, 20], "score": 0.9},
           {"image_id": 1, "category_id": 1, "bbox": [52, 50, 40, 40], "score": 0.8},
           {"image_id": 1, "category_id": 1, "bbox": [200, 200, 10, 10], "score": 0.7}]
    st = CE.evaluate_bbox(gt, res)["stats"]
    assert st["AP50"] == pytest.approx(1.0) and st["AP75"] == pytest.approx(1.0)  # both found before the stray
    assert st["AR1"] == pytest.approx(0.5) and st["AR100"] == pytest.approx(0.95)  # the shifted box fails only IoU 0.95
    assert st["APl"] == -1.0  # no large instance
    assert CE.evaluate_bbox(gt, [])["stats"]["AP"] == pytest.approx(0.0)


def test_panoptic_quality_hand_cases():
    """evaluation/pq.py (restated from the PQ paper / panopticapi; parity unpinned - the reference has neither the package nor
    a golden for it) on cases small enough to do by hand."""
    from u2seg_amd.evaluation.pq import pq_compute_arrays

    cats = {1: {"isthing": 1}, 2: {"isthing": 1}, 7: {"isthing": 0}}
    gt = np.zeros((10, 10), dtype=np.int64)
    gt[:5, :5] = 1     # thing, category 1, 25 px
    gt[5:, :] = 2      # stuff, category 7, 50 px
    gt[:5, 5:8] = 3    # thing, category 2, 15 px; the 10 px of columns 8-9 stay void
    gt_segs = [{"id": 1, "category_id": 1, "iscrowd": 0}, {"id": 2, "category_id": 7, "iscrowd": 0},
               {"id": 3, "category_id": 2, "iscrowd": 0}]
    # perfect prediction (different ids): PQ = SQ = RQ = 1 everywhere
    pred = gt * 10
    segs = [{"id": 10, "category_id": 1}, {"id": 20, "category_id": 7}, {"id": 30, "category_id": 2}]
    res = pq_compute_arrays([(gt, gt_segs, pred, segs)], cats)
    assert res["All"] == {"pq": 1.0, "sq": 1.0, "rq": 1.0, "n": 3} and res["Things"]["n"] == 2 and res["Stuff"]["n"] == 1
    # segment 10 shrinks to 20 of its 25 px (IoU 0.8); segment 30 gets the wrong category (one FP + one FN, categories 1 / 2);
    # a stray segment lies entirely on void pixels (excused); the stuff segment spills 4 px onto void (they do not count)
    pred = np.zeros_like(gt)
    pred[:4, :5] = 10
    pred[5:, :] = 20
    pred[:5, 5:8] = 30
    pred[:2, 8:] = 40
    pred[2:4, 8:] = 20
    segs = [{"id": 10, "category_id": 1}, {"id": 20, "category_id": 7}, {"id": 30, "category_id": 1}, {"id": 40, "category_id": 2}]
    res = pq_compute_arrays([(gt, gt_segs, pred, segs)], cats)
    pc = res["per_class"]
    assert pc[1]["sq"] == pytest.approx(0.8) and pc[1]["rq"] == pytest.approx(1 / 1.5) and pc[1]["pq"] == pytest.approx(0.8 / 1.5)
    assert pc[2] == {"pq": 0.0, "sq": 0.0, "rq": 0.0}   # its only instance was missed, the stray on void is not an FP
    assert pc[7]["pq"] == pytest.approx(1.0)              # 50 / (54 + 50 - 50 - 4)
    assert res["All"]["pq"] == pytest.approx((0.8 / 1.5 + 0 + 1.0) / 3) and res["Stuff"]["pq"] == pytest.approx(1.0)
    # crowd ground truth: never a TP or FN, and a prediction of its category lying on it is not an FP
    gt_segs[2]["iscrowd"] = 1
    segs = [{"id": 10, "category_id": 1}, {"id": 20, "category_id": 7}, {"id": 30, "category_id": 2}, {"id": 40, "category_id": 2}]
    res = pq_compute_arrays([(gt, gt_segs, pred, segs)], cats)
    assert res["per_class"][2] == {"pq": 0.0, "sq": 0.0, "rq": 0.0} and res["Things"]["n"] == 1
    with pytest.raises(KeyError):
        pq_compute_arrays([(gt, gt_segs, pred, segs[:2])], cats)  # an id in the png without segments_info


def test_panoptic_quality_file_form(tmp_path):
    """pq.pq_compute on json + png files (panopticapi's calling convention) == the array form."""
    from u2seg_amd.data.pseudo_panoptic import id2rgb
    from u2seg_amd.evaluation.pq import pq_compute, pq_compute_arrays

    cats = [{"id": 1, "isthing": 1}, {"id": 7, "isthing": 0}]
    rs = np.random.RandomState(3)
    samples, gt_json, pred_json = [], {"categories": cats, "annotations": []}, {"annotations": []}
    os.makedirs(tmp_path / "gt")
    os.makedirs(tmp_path / "pred")
    for i in range(3):
        gt = rs.randint(0, 4, (6, 8)).repeat(4, 0).repeat(4, 1)
        pred = np.roll(gt, 2, axis=1) * 1000  # ids above 255: exercises the 3-byte id encoding
        gsegs = [{"id": int(k), "category_id": 1 if k < 3 else 7, "iscrowd": 0} for k in np.unique(gt) if k]
        psegs = [{"id": int(k), "category_id": 1 if k < 3000 else 7} for k in np.unique(pred) if k]
        name = "%d.png" % i
        Image.fromarray(id2rgb(gt)).save(tmp_path / "gt" / name)
        Image.fromarray(id2rgb(pred)).save(tmp_path / "pred" / name)
        gt_json["annotations"].append({"image_id": i, "file_name": name, "segments_info": gsegs})
        pred_json["annotations"].append({"image_id": i, "file_name": name, "segments_info": psegs})
        samples.append((gt, gsegs, pred, psegs))
    json.dump(gt_json, open(tmp_path / "gt.json", "w"))
    json.dump(pred_json, open(tmp_path / "pred.json", "w"))
    a = pq_compute(str(tmp_path / "gt.json"), str(tmp_path / "pred.json"), str(tmp_path / "gt"), str(tmp_path / "pred"))
    b = pq_compute_arrays(samples, {c["id"]: c for c in cats})
    assert a == b and 0 < a["All"]["pq"] < 1
