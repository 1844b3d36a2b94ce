#!/usr/bin/env python
"""Input-pipeline throughput: DatasetMapper (JPEG decode, ResizeShortestEdge over the config's 10 short-edge choices, flip,
RLE -> bitmasks) in DataLoader workers, optionally through DevicePrefetcher into HBM.  Builds a throw-away COCO-shaped
dataset (480x640 / 640x480 JPEGs, 7 RLE instances per image, semantic label maps) under a temp dir, so it runs anywhere.
Prints one JSON line per worker count; the training step consumes ~197 img/s per GPU."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_dataset(root, n_images, seed=0):
    from PIL import Image

    from u2seg_amd.data import rle

    rs = np.random.RandomState(seed)
    img_dir = os.path.join(root, "coco", "train2017")
    ann_dir = os.path.join(root, "prepare_ours", "u2seg_annotations", "ins_annotations")
    sem_dir = os.path.join(root, "prepare_ours", "u2seg_annotations", "panoptic_annotations", "panoptic_stuff_cocotrain_800")
    for d in (img_dir, ann_dir, sem_dir):
        os.makedirs(d, exist_ok=True)
    images, annotations, aid = [], [], 1
    for i in range(n_images):
        h, w = (480, 640) if i % 4 else (640, 480)
        name = "%012d" % (i + 1)
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.clip(np.stack([xx * 255.0 / w, yy * 255.0 / h, (xx + yy) * 255.0 / (h + w)], 2) + rs.randn(h, w, 3) * 20, 0, 255)
        Image.fromarray(img.astype(np.uint8)).save(os.path.join(img_dir, name + ".jpg"), quality=90)
        sem = rs.randint(0, 28, (h // 32 + 1, w // 32 + 1)).repeat(32, 0).repeat(32, 1)[:h, :w].astype(np.uint8)
        Image.fromarray(sem, mode="L").save(os.path.join(sem_dir, name + ".png"))
        images.append({"id": i + 1, "file_name": name + ".jpg", "height": h, "width": w})
        for _ in range(7):
            bw, bh = rs.uniform(0.1, 0.5) * w, rs.uniform(0.1, 0.5) * h
            x0, y0 = rs.uniform(0, w - bw), rs.uniform(0, h - bh)
            m = (((xx + 0.5 - x0 - bw / 2) / (bw / 2)) ** 2 + ((yy + 0.5 - y0 - bh / 2) / (bh / 2)) ** 2 <= 1).astype(np.uint8)
            annotations.append({"id": aid, "image_id": i + 1, "category_id": int(rs.randint(1, 801)), "iscrowd": 0,
                                "bbox": [float(x0), float(y0), float(bw), float(bh)], "area": float(m.sum()),
                                "segmentation": rle.encode(m)})
            aid += 1
    cats = [{"id": c + 1, "name": str(c + 1), "supercategory": str(c + 1)} for c in range(800)]
    json.dump({"images": images, "annotations": annotations, "categories": cats},
              open(os.path.join(ann_dir, "cocotrain_800.json"), "w"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=48)
    ap.add_argument("--workers", type=int, nargs="+", default=[0, 8, 32])
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--device", default="cuda:0" if torch.cuda.is_available() else "cpu")
    a = ap.parse_args()
    os.environ["CLUSTER_NUM"] = "800"
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import DevicePrefetcher, build_detection_train_loader, register_all_coco

    root = tempfile.mkdtemp(prefix="u2seg_bench_data_")
    t0 = time.time()
    make_dataset(root, a.images)
    register_all_coco(root)
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml"))
    print(json.dumps({"dataset": "%d synthetic JPEGs 480x640 / 640x480, 7 RLE instances each" % a.images,
                      "build_s": round(time.time() - t0, 1), "short_edges": list(cfg.INPUT.MIN_SIZE_TRAIN)}))
    for nw in a.workers:
        cfg.defrost()
        cfg.merge_from_list(["DATALOADER.NUM_WORKERS", nw])
        loader = build_detection_train_loader(cfg, seed=1)
        pre = DevicePrefetcher(loader, a.device) if a.device.startswith("cuda") else None
        stream = iter(pre if pre is not None else loader)
        for _ in range(a.warmup):  # worker start-up, pinned staging blocks, and the batches the workers prefetched
            next(stream)
        if pre is not None:
            pre.seconds_waiting_for_loader = pre.seconds_staging = 0.0
        t0, n, px = time.time(), 0, 0
        for _ in range(a.batches):
            batch = next(stream)
            n += len(batch)
            px += sum(d["image"].shape[1] * d["image"].shape[2] for d in batch)
        if a.device.startswith("cuda"):
            torch.cuda.synchronize()
        dt = time.time() - t0
        print(json.dumps({"metric": "input pipeline img/s", "workers": nw, "batch": len(batch), "img_per_s": round(n / dt, 1),
                          "mean_megapixels": round(px / n / 1e6, 2), "to_device": a.device,
                          "host_s_waiting_for_loader": round(pre.seconds_waiting_for_loader, 2) if pre else None,
                          "host_s_staging": round(pre.seconds_staging, 2) if pre else None, "wall_s": round(dt, 2),
                          "host_threads": torch.get_num_threads()}))
        del stream, loader, pre


if __name__ == "__main__":
    main()
