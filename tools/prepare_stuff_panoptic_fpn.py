#!/usr/bin/env python
"""Command line of the reference's datasets/prepare_ours/prepare_stuff_panoptic_fpn.py (--split, --cluster_num), run from
the directory that holds ./datasets: semantic label maps for the Panoptic-FPN semantic head from the pseudo panoptic pngs."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from u2seg_amd.data.pseudo_panoptic import separate_semantic_from_panoptic  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--split", type=str, default="train")
    ap.add_argument("--cluster_num", type=str, default="800")
    a = ap.parse_args()
    base = "datasets/prepare_ours/u2seg_annotations/panoptic_annotations"
    js = "%s/coco%s_%s.json" % (base, a.split, a.cluster_num)
    n = separate_semantic_from_panoptic(js, "%s/coco%s_%s" % (base, a.split, a.cluster_num),
                                        "%s/panoptic_stuff_coco%s_%s" % (base, a.split, a.cluster_num),
                                        json.load(open(js))["categories"])
    print("wrote %d label maps" % n)
