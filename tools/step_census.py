#!/usr/bin/env python
"""Launch census of ONE training step out of a rocprofv3 --kernel-trace CSV: the dispatches between two consecutive optimizer
launches (sgd_clip kernels), i.e. without the warm-up steps' first-use launches that the --stats totals / steps figure includes.
usage: tools/step_census.py <dir with *_kernel_trace.csv> [top]"""
import collections
import csv
import glob
import sys


def short(n):
    for junk in ("void ", "(anonymous namespace)::", "at::native::", "u2conv::"):
        n = n.replace(junk, "")
    return n[:90]


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if "sgd" in r["Kernel_Name"].lower()]
    if len(opt) < 2:
        print("fewer than two optimizer launches in the trace")
        return
    seg = rows[opt[-2] + 1:opt[-1] + 1]
    fam = {"conv forward / data gradient": ("conv_halo", "conv_tile", "conv_stream", "conv_igemm"),
           "weight gradients (+ fused backward)": ("wgrad",),
           "normalisation family": ("colreduce", "norm_bwd", "affine_act", "bn_act", "bn_bwd", "bn_finalize", "gn_finalize", "affine_upadd", "relu_bwd", "add_n", "upadd"),
           "ROIAlign": ("roi_",), "ATen": ("at::", "elementwise_kernel", "reduce_kernel", "Cat", "gather", "radixSort", "index"),
           "copies": ("copyBuffer",)}
    acc = collections.OrderedDict((k, [0, 0.0]) for k in list(fam) + ["other u2seg kernels"])
    per = collections.Counter()
    dur = collections.Counter()
    for r in seg:
        name = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        for k, pats in fam.items():
            if any(p in name for p in pats):
                break
        else:
            k = "other u2seg kernels"
        acc[k][0] += 1
        acc[k][1] += d
        per[short(name)] += 1
        dur[short(name)] += d
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    print("one step = the %d dispatches between two optimizer launches: %.2f ms of kernel time, %.2f ms from first start to last end"
          % (len(seg), sum(v[1] for v in acc.values()), span))
    for k, (n, d) in acc.items():
        print("  %-38s %5d launches  %7.2f ms" % (k, n, d))
    print("by kernel (launches, ms):")
    for name, n in per.most_common(top):
        print("  %4d  %7.3f  %s" % (n, dur[name], name))


if __name__ == "__main__":
    main()
