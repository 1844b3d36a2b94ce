"""Pseudo panoptic labels = class-aware pseudo instance masks painted over an unsupervised semantic map
(datasets/prepare_ours/generate_pseudo_panoptic.py:43-173; the reference is a script, this is the same procedure as
functions plus `tools/generate_pseudo_panoptic.py` with its command line and directory layout).

Per image: instances are painted in order of decreasing box area, each with the next free segment id (ids run on across
images); instances that end up completely covered are dropped; every semantic class 1..27 claims the pixels no instance
took unless more than 70 % of the class already lies under instances; the id map is written as an RGB png
(id = R + 256 G + 256^2 B, panopticapi.utils.id2rgb)."""
import json
import os

import numpy as np
from PIL import Image

from . import rle

NUM_STUFF = 27
COVERED_FRACTION = 0.7


def create_cate(num):
    """Category table of the merged annotation: ids 1..num are things, the following 27 are stuff (:13-25)."""
    return [{"supercategory": str(i + 1), "id": i + 1, "name": str(i + 1), "isthing": 1 if i + 1 <= num else 0}
            for i in range(num + NUM_STUFF)]


def id2rgb(id_map):
    out = np.zeros(id_map.shape + (3,), dtype=np.uint8)
    rest = id_map.astype(np.uint32).copy()
    for c in range(3):
        out[..., c] = rest % 256
        rest //= 256
    return out


def rgb2id(color):
    c = np.asarray(color, dtype=np.uint32)
    return c[..., 0] + 256 * c[..., 1] + 256 * 256 * c[..., 2]


def merge_image(semantic, instances, class_num, first_id):
    """semantic: int array [h, w] with labels 0..26 (the unsupervised segmenter's output); instances: the image's pseudo
    instance records ({"bbox": [x, y, w, h], "segmentation": RLE, "category_id", ...}; they receive an "id").
    Returns (id map uint32 [h, w], segments_info, next free id)."""
    sem = np.asarray(semantic) + 1
    combined = np.zeros(sem.shape, dtype=np.uint32)
    masks = [rle.decode(ins["segmentation"]) for ins in instances]
    order = sorted(range(len(instances)), key=lambda k: instances[k]["bbox"][-2] * instances[k]["bbox"][-1], reverse=True)
    seg_id, painted = first_id, []
    for k in order:
        combined[masks[k] == 1] = seg_id
        instances[k]["id"] = seg_id
        painted.append(instances[k])
        seg_id += 1
    present = set(np.unique(combined).tolist())
    segments = [ins for ins in painted if ins["id"] in present]  # later, smaller instances may have covered one entirely
    for cat in range(1, NUM_STUFF + 1):
        is_cat = sem == cat
        free = is_cat & (combined == 0)
        if not free.any():
            continue
        if np.sum(is_cat & (combined != 0)) / np.sum(is_cat) > COVERED_FRACTION:
            continue
        combined[free] = seg_id
        segments.append({"category_id": cat + class_num, "id": seg_id, "iscrowd": 0, "bbox": [], "area": 0})
        seg_id += 1
    return combined, segments, seg_id


def semantic_file_table(names_file):
    """image file name with .png extension -> '<line number>.npy' (:55-60: the i-th line names the image whose semantic
    map is stored as i.npy)."""
    table = {}
    with open(names_file) as f:
        for i, line in enumerate(f):
            table[line[:-4] + "png"] = "%d.npy" % i
    return table


def generate(root, class_num=800, split="train"):
    """The whole script over the reference's directory layout below `root` (its paths are relative to the working
    directory): reads the panoptic template, the pseudo instance file and the semantic maps, writes the png id maps and
    coco{split}_{class_num}.json under prepare_ours/u2seg_annotations/panoptic_annotations/.  Returns the json dict."""
    ann_root = os.path.join(root, "datasets", "prepare_ours", "u2seg_annotations")
    template = json.load(open(os.path.join(root, "datasets", "datasets", "panoptic_anns", "panoptic_%s2017.json" % split)))
    pseudo = json.load(open(os.path.join(ann_root, "ins_annotations", "coco%s_%d_ins_panoptic.json" % (split, class_num))))
    table = semantic_file_table(os.path.join(ann_root, "semantic_annotations", "coco_%s_img_file_names.txt" % split))
    sem_dir = os.path.join(ann_root, "semantic_annotations", "stego_coco_%s_semantic_seg_resized" % split)
    save_root = os.path.join(ann_root, "panoptic_annotations", "coco%s_%d" % (split, class_num))
    out = {"images": template["images"], "info": template["info"], "licenses": template["licenses"], "annotations": [],
           "categories": create_cate(class_num)}
    seen = {img["id"]: False for img in template["images"]}
    seg_id = 1
    for ann in template["annotations"]:
        semantic = np.load(os.path.join(sem_dir, table[ann["file_name"]]))
        record = pseudo["annotations"].get(str(ann["image_id"]))
        if record is None:
            continue  # an image without pseudo instances is left out altogether
        combined, segments, seg_id = merge_image(semantic, record["segments_info"], class_num, seg_id)
        seen[ann["image_id"]] = True
        os.makedirs(save_root, exist_ok=True)
        Image.fromarray(id2rgb(combined)).save(os.path.join(save_root, ann["file_name"]))
        out["annotations"].append({"file_name": ann["file_name"], "image_id": ann["image_id"], "segments_info": segments})
    out["images"] = [img for img in out["images"] if seen[img["id"]]]
    with open(os.path.join(ann_root, "panoptic_annotations", "coco%s_%d.json" % (split, class_num)), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False)
    return out


# ------------------------------------------------------------------------------------------------
# the other label-preparation steps of datasets/prepare_ours/
# ------------------------------------------------------------------------------------------------
def classaware_instance_annotations(template, cluster_results, mask_annotations, num_categories=300):
    """generate_classaware_instanceseg_annotations.py:36-70: class-agnostic instance masks (CutLER / MaskCut records with
    an "ins_id") receive the cluster id of their crop ("<ins_id>.jpg" in the clustering result) as category and their
    ins_id as annotation id; images without any instance are dropped.  Returns the COCO-format dict."""
    out = {"licenses": template["licenses"],
           "categories": [{"id": i + 1, "name": str(i + 1), "supercategory": str(i + 1)} for i in range(num_categories)],
           "images": template["images"], "info": template["info"], "annotations": []}
    seen = set()
    for ann in mask_annotations:
        ann["category_id"] = cluster_results[str(ann["ins_id"]) + ".jpg"]
        ann["id"] = ann["ins_id"]
        out["annotations"].append(ann)
        seen.add(ann["image_id"])
    out["images"] = [img for img in out["images"] if img["id"] in seen]
    return out


# COCO panoptic stuff category id -> 1-based index of its supercategory (get_panoptic_anns_supercategory.py:9-13); the
# contiguous-order form of the same table is evaluation/sem_seg_evaluation.py:STUFF_TO_SUPERCATEGORY
STUFF_ID_TO_SUPERCATEGORY = dict(zip(
    (92, 93, 95, 100, 107, 109, 112, 118, 119, 122, 125, 128, 130, 133, 138, 141, 144, 145, 147, 148, 149, 151, 154, 155, 156,
     159, 161, 166, 168, 171, 175, 176, 177, 178, 180, 181, 184, 185, 186, 187, 188, 189, 190, 191, 192, 193, 194, 195, 196,
     197, 198, 199, 200),
    (1, 1, 2, 3, 4, 1, 4, 5, 6, 7, 8, 2, 4, 4, 9, 1, 8, 8, 8, 10, 8, 2, 8, 10, 4, 8, 4, 2, 1, 11, 11, 11, 11, 10, 12, 12, 6, 9,
     13, 14, 4, 4, 5, 8, 15, 6, 8, 3, 7, 2, 15, 11, 1)))


def panoptic_supercategory_annotations(standard, cluster_num):
    """get_panoptic_anns_supercategory.py:15-27: in the ground-truth panoptic json every stuff category id becomes
    cluster_num + its supercategory index, in the segments and in the category table (in place; returns `standard`)."""
    for ann in standard["annotations"]:
        for seg in ann["segments_info"]:
            if seg["category_id"] in STUFF_ID_TO_SUPERCATEGORY:
                seg["category_id"] = STUFF_ID_TO_SUPERCATEGORY[seg["category_id"]] + cluster_num
    for cate in standard["categories"]:
        if cate["id"] in STUFF_ID_TO_SUPERCATEGORY:
            cate["id"] = STUFF_ID_TO_SUPERCATEGORY[cate["id"]] + cluster_num
    return standard


def separate_semantic_from_panoptic(panoptic_json, panoptic_root, sem_seg_root, categories):
    """prepare_stuff_panoptic_fpn.py:21-75: one uint8 label map per panoptic png - things 0, the stuff categories 1.. in the
    order of `categories`, everything unlabelled 255 - the `sem_seg_file_name` inputs of the training data path."""
    os.makedirs(sem_seg_root, exist_ok=True)
    stuff_ids = [k["id"] for k in categories if k["isthing"] == 0]
    assert len(stuff_ids) <= 254
    id_map = {sid: i + 1 for i, sid in enumerate(stuff_ids)}
    id_map.update({k["id"]: 0 for k in categories if k["isthing"] == 1})
    id_map[0] = 255
    with open(panoptic_json) as f:
        obj = json.load(f)
    for anno in obj["annotations"]:
        panoptic = rgb2id(np.asarray(Image.open(os.path.join(panoptic_root, anno["file_name"])), dtype=np.uint32))
        output = np.full(panoptic.shape, 255, dtype=np.uint8)
        for seg in anno["segments_info"]:
            output[panoptic == seg["id"]] = id_map[seg["category_id"]]
        Image.fromarray(output).save(os.path.join(sem_seg_root, anno["file_name"]))
    return len(obj["annotations"])
