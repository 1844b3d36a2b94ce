"""COCO-format dataset loading and the registrations the U2Seg configs name
(detectron2/data/datasets/coco.py:36-232 and 235-312, coco_panoptic.py:102-198, builtin.py:33-165, builtin_meta.py:14-39
and 277-322).  pycocotools' COCO index is replaced by a direct pass over the json (same ordering rules: images by id,
annotations in file order per image, categories by id)."""
import copy
import json
import os
from collections import defaultdict

from . import rle
from .catalog import DatasetCatalog, MetadataCatalog
from .detection_utils import BoxMode


def cluster_num():
    """The reference fixes the number of pseudo-classes at import time from the environment (builtin.py:33)."""
    return int(os.getenv("CLUSTER_NUM", "800"))


def u2seg_categories(num):
    """builtin_meta.py:14-39 (create_cate) without the random colours: ids 1..num are things, the next 27 are stuff."""
    return [{"supercategory": str(i + 1), "id": i + 1, "name": str(i + 1), "isthing": 1 if i + 1 <= num else 0}
            for i in range(num + 27)]


def instances_meta(num=None):
    cats = u2seg_categories(cluster_num() if num is None else num)
    thing_ids = [k["id"] for k in cats if k["isthing"] == 1]
    return {"thing_dataset_id_to_contiguous_id": {k: i for i, k in enumerate(thing_ids)},
            "thing_classes": [k["name"] for k in cats if k["isthing"] == 1]}


def panoptic_separated_meta(num=None):
    cats = u2seg_categories(cluster_num() if num is None else num)
    stuff_ids = [k["id"] for k in cats if k["isthing"] == 0]
    ids = {k: i + 1 for i, k in enumerate(stuff_ids)}
    ids[0] = 0  # "things" collapse to label 0 in the semantic maps
    ret = {"stuff_dataset_id_to_contiguous_id": ids,
           "stuff_classes": ["things"] + [k["name"] for k in cats if k["isthing"] == 0]}
    ret.update(instances_meta(num))
    return ret


def load_coco_json(json_file, image_root, dataset_name=None, extra_annotation_keys=None):
    """list of {"file_name", "height", "width", "image_id", "annotations": [{"iscrowd", "bbox", "category_id",
    "segmentation", "bbox_mode": XYWH_ABS}]}; category ids mapped to 0..C-1 when `dataset_name` is given."""
    with open(json_file) as f:
        data = json.load(f)
    id_map = None
    if dataset_name is not None:
        cats = sorted(data.get("categories", []), key=lambda c: c["id"])
        cat_ids = [c["id"] for c in cats]
        meta = MetadataCatalog.get(dataset_name)
        meta.thing_classes = [c["name"] for c in cats]
        id_map = {v: i for i, v in enumerate(cat_ids)}
        meta.thing_dataset_id_to_contiguous_id = id_map
    images = {im["id"]: im for im in data["images"]}
    per_image = defaultdict(list)
    for ann in data.get("annotations", []):
        if ann["image_id"] in images:
            per_image[ann["image_id"]].append(ann)
    if "minival" not in json_file:
        ann_ids = [a["id"] for anns in per_image.values() for a in anns]
        assert len(set(ann_ids)) == len(ann_ids), "Annotation ids in '{}' are not unique!".format(json_file)
    ann_keys = ["iscrowd", "bbox", "keypoints", "category_id"] + (extra_annotation_keys or [])
    dataset_dicts = []
    for img_id in sorted(images):
        img = images[img_id]
        record = {"file_name": os.path.join(image_root, img["file_name"]), "height": img["height"], "width": img["width"],
                  "image_id": img["id"]}
        objs = []
        for anno in per_image.get(img_id, []):
            assert anno.get("ignore", 0) == 0, '"ignore" in COCO json file is not supported.'
            obj = {key: anno[key] for key in ann_keys if key in anno}
            if "bbox" in obj and len(obj["bbox"]) == 0:
                raise ValueError("One annotation of image {} contains empty 'bbox' value! This json does not have valid "
                                 "COCO format.".format(img_id))
            segm = anno.get("segmentation", None)
            if segm:
                if isinstance(segm, dict):
                    if isinstance(segm["counts"], list):
                        segm = rle.compress(segm)
                else:
                    segm = [poly for poly in segm if len(poly) % 2 == 0 and len(poly) >= 6]
                    if len(segm) == 0:
                        continue  # an instance without a valid polygon is dropped
                obj["segmentation"] = segm
            obj["bbox_mode"] = BoxMode.XYWH_ABS
            if id_map:
                try:
                    obj["category_id"] = id_map[obj["category_id"]]
                except KeyError as e:
                    raise KeyError("Encountered category_id={} but this id does not exist in 'categories' of the json "
                                   "file.".format(obj["category_id"])) from e
            objs.append(obj)
        record["annotations"] = objs
        dataset_dicts.append(record)
    return dataset_dicts


def load_sem_seg(gt_root, image_root, gt_ext="png", image_ext="jpg"):
    """Pairs every image under image_root with the label map of the same stem under gt_root (coco.py:235-312)."""
    def stem(folder, path):
        return os.path.splitext(os.path.normpath(os.path.relpath(path, start=folder)))[0]

    input_files = sorted((os.path.join(image_root, f) for f in os.listdir(image_root) if f.endswith(image_ext)),
                         key=lambda p: stem(image_root, p))
    gt_files = sorted((os.path.join(gt_root, f) for f in os.listdir(gt_root) if f.endswith(gt_ext)),
                      key=lambda p: stem(gt_root, p))
    assert len(gt_files) > 0, "No annotations found in {}.".format(gt_root)
    if len(input_files) != len(gt_files):
        common = sorted({os.path.basename(f)[: -len(image_ext)] for f in input_files}
                        & {os.path.basename(f)[: -len(gt_ext)] for f in gt_files})
        input_files = [os.path.join(image_root, f + image_ext) for f in common]
        gt_files = [os.path.join(gt_root, f + gt_ext) for f in common]
    return [{"file_name": i, "sem_seg_file_name": g} for i, g in zip(input_files, gt_files)]


def merge_to_panoptic(detection_dicts, sem_seg_dicts):
    by_file = {x["file_name"]: x for x in sem_seg_dicts}
    assert len(by_file) > 0
    results = []
    for det in detection_dicts:
        dic = copy.copy(det)
        dic.update(by_file[dic["file_name"]])
        results.append(dic)
    return results


def register_coco_instances(name, metadata, json_file, image_root):
    DatasetCatalog.register(name, lambda: load_coco_json(json_file, image_root, name))
    MetadataCatalog.get(name).set(json_file=json_file, image_root=image_root, evaluator_type="coco", **metadata)


def register_coco_panoptic_separated(name, metadata, image_root, panoptic_root, panoptic_json, sem_seg_root,
                                     instances_json):
    """`name + "_separated"`: instance annotations + semantic label maps per image; `name + "_stuffonly"`: label maps."""
    panoptic_name = name + "_separated"
    DatasetCatalog.register(panoptic_name, lambda: merge_to_panoptic(
        load_coco_json(instances_json, image_root, panoptic_name), load_sem_seg(sem_seg_root, image_root)))
    MetadataCatalog.get(panoptic_name).set(panoptic_root=panoptic_root, image_root=image_root, panoptic_json=panoptic_json,
                                           sem_seg_root=sem_seg_root, json_file=instances_json,
                                           evaluator_type="coco_panoptic_seg", ignore_label=255, **metadata)
    semantic_name = name + "_stuffonly"
    DatasetCatalog.register(semantic_name, lambda: load_sem_seg(sem_seg_root, image_root))
    MetadataCatalog.get(semantic_name).set(sem_seg_root=sem_seg_root, image_root=image_root, evaluator_type="sem_seg",
                                           ignore_label=255, **metadata)


def builtin_splits(num=None):
    """The U2Seg entries of builtin.py:59-118: (instances, panoptic) path tables keyed by dataset name."""
    n = cluster_num() if num is None else num
    coco = {
        "coco_2017_train": ("./coco/train2017", "./prepare_ours/u2seg_annotations/ins_annotations/cocotrain_%d.json" % n),
        "coco_2017_val": ("./coco/val2017", "./coco/annotations/instances_val2017.json"),
    }
    panoptic = {
        "coco_2017_train_panoptic": (
            "./prepare_ours/u2seg_annotations/panoptic_annotations/cocotrain_%d" % n,
            "./prepare_ours/u2seg_annotations/panoptic_annotations/cocotrain_%d.json" % n,
            "./prepare_ours/u2seg_annotations/panoptic_annotations/panoptic_stuff_cocotrain_%d" % n),
        "coco_2017_val_panoptic": (
            "datasets/panoptic_anns/panoptic_val2017",
            "datasets/panoptic_anns/panoptic_val2017_%dsuper.json" % n,
            "datasets/panoptic_anns/panoptic_stuff_val2017"),
    }
    return coco, panoptic


def register_all_coco(root=None, num=None):
    """Registers coco_2017_{train,val} and coco_2017_{train,val}_panoptic_{separated,stuffonly} under `root`.  The fork
    hard-wires the root to ./datasets (builtin.py:277-281; upstream's DETECTRON2_DATASETS lookup is commented out there) -
    that stays the default, and the environment variable is honoured when set.  Loading stays lazy: nothing is read until
    DatasetCatalog.get."""
    root = os.path.expanduser(os.getenv("DETECTRON2_DATASETS", "datasets")) if root is None else root
    coco, panoptic = builtin_splits(num)
    for key, (image_root, json_file) in coco.items():
        if key not in DatasetCatalog:
            register_coco_instances(key, instances_meta(num), os.path.join(root, json_file), os.path.join(root, image_root))
    for prefix, (panoptic_root, panoptic_json, semantic_root) in panoptic.items():
        if prefix + "_separated" in DatasetCatalog:
            continue
        inst = MetadataCatalog.get(prefix[: -len("_panoptic")])
        register_coco_panoptic_separated(prefix, panoptic_separated_meta(num), inst.image_root,
                                         os.path.join(root, panoptic_root), os.path.join(root, panoptic_json),
                                         os.path.join(root, semantic_root), inst.json_file)
