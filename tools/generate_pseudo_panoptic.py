#!/usr/bin/env python
"""Command line of the reference's datasets/prepare_ours/generate_pseudo_panoptic.py (--class_num, --split), run from the
directory that holds ./datasets like the reference."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from u2seg_amd.data.pseudo_panoptic import generate  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-class_num", "--class_num", type=int, default=800)
    ap.add_argument("-split", "--split", type=str, default="train")
    a = ap.parse_args()
    out = generate(os.getcwd(), a.class_num, a.split)
    print("wrote %d images, %d annotations" % (len(out["images"]), len(out["annotations"])))
