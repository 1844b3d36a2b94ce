#!/usr/bin/env python
"""Where the chip is idle inside ONE overlapped training step, out of a rocprofv3 --kernel-trace CSV: the union of the kernels'
[start, end] intervals between two consecutive optimizer launches against the step's span, and the longest gaps with the kernels on
either side of them (a gap = no kernel of ANY stream running: a host synchronisation, a launch-bound stretch, a dependency chain).
usage: tools/step_idle.py <dir with *_kernel_trace.csv> [top] [from_ms to_ms]   (a window: every dispatch in it, with its queue)"""
import csv
import glob
import sys

from step_census import short


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
    if len(opt) < 3:
        print("fewer than three optimizer launches in the trace")
        return
    seg = rows[opt[-3] + 1:opt[-2] + 1]   # the step before the last one (bench.py's last timed step is the serial roofline sample)
    t0 = int(seg[0]["Start_Timestamp"])
    span = (int(seg[-1]["End_Timestamp"]) - t0) / 1e6
    busy, gaps = 0.0, []
    cur_s, cur_e, last = int(seg[0]["Start_Timestamp"]), int(seg[0]["End_Timestamp"]), seg[0]
    for r in seg[1:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(((s - cur_e) / 1e3, (cur_e - t0) / 1e6, short(last["Kernel_Name"]), short(r["Kernel_Name"])))
            cur_s, cur_e, last = s, e, r
        elif e > cur_e:
            cur_e, last = e, r
    busy += cur_e - cur_s
    busy /= 1e6
    ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
    print("one step = %d dispatches: span %.2f ms, some kernel running %.2f ms, idle %.2f ms in %d gaps; sum of kernel durations %.2f ms"
          % (len(seg), span, busy, span - busy, len(gaps), ksum))
    hist = [(1, 0, 0.0), (5, 0, 0.0), (20, 0, 0.0), (100, 0, 0.0), (1e9, 0, 0.0)]
    hist = [list(h) for h in hist]
    for g in gaps:
        for h in hist:
            if g[0] < h[0]:
                h[1] += 1
                h[2] += g[0]
                break
    print("gaps by length (us): " + ", ".join("<%g: %d (%.2f ms)" % (h[0], h[1], h[2] / 1e3) for h in hist))
    bins = {}
    for g in gaps:
        bins[int(g[1])] = bins.get(int(g[1]), 0.0) + g[0]
    print("idle us per ms of the step: " + " ".join("%d:%d" % (b, bins.get(b, 0)) for b in range(int(span) + 1)))
    print("longest gaps (us, at ms of the step, kernel before -> kernel after):")
    for g in sorted(gaps, reverse=True)[:top]:
        print("  %7.1f  @%6.2f  %s  ->  %s" % (g[0], g[1], g[2][:60], g[3][:60]))
    if len(sys.argv) > 4:
        lo, hi = float(sys.argv[3]), float(sys.argv[4])
        queues = {}
        print("dispatches starting in [%.2f, %.2f) ms: start ms, duration us, queue, kernel" % (lo, hi))
        for r in seg:
            st = (int(r["Start_Timestamp"]) - t0) / 1e6
            if lo <= st < hi:
                q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
                print("  %7.3f %8.1f  q%d  %s" % (st, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, q, short(r["Kernel_Name"])[:70]))


if __name__ == "__main__":
    main()
