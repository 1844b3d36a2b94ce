#!/usr/bin/env python
"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, separate passes) of
`python bench.py --steps 1 --warmup 1 --no-cpu-baseline`: HBM bytes per launch of the two MFMA kernel families.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<note>"

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): the counters are in KiB; on gfx950 FETCH_SIZE
reports half of a wide coalesced streaming read and is doubled; WRITE_SIZE is uncalibrated and taken as is."""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(d, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r.get("Kernel_Name", "")
            agg[k][0] += 1
            agg[k][1] += float(r.get("Counter_Value", 0) or 0)
    return agg


def family(name):
    if "wgrad_stream_kernel" in name:  # the 1x1 streaming kernel; its DG instantiations carry a data gradient as well
        return "conv_wgrad"
    if "conv_wgrad" in name or "wgrad_reduce_kernel" in name:  # the reduction pass belongs to the launch whose partial tiles it sums
        return "conv_wgrad"
    if "conv_tile_kernel" in name or "conv_igemm" in name or "conv_halo_kernel" in name or "conv_stream_kernel" in name:
        return "conv_igemm"
    return None


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for fam in ("conv_igemm", "conv_wgrad"):
        nf = sum(v[0] for k, v in fetch.items() if family(k) == fam and "wgrad_reduce_kernel" not in k)
        fb = sum(v[1] for k, v in fetch.items() if family(k) == fam) * 1024 * 2
        nw = sum(v[0] for k, v in write.items() if family(k) == fam and "wgrad_reduce_kernel" not in k)
        wb = sum(v[1] for k, v in write.items() if family(k) == fam) * 1024
        if nf and nw:
            out[fam] = {"launches": nf, "fetch_bytes_per_launch": fb / nf, "write_bytes_per_launch": wb / nw,
                        "hbm_bytes_per_launch": fb / nf + wb / nw}
    out["note"] = sys.argv[4]
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
