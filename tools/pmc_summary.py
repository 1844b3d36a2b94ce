#!/usr/bin/env python
"""Summarises a rocprofv3 --pmc counter_collection CSV: per kernel, launches and mean counter value per launch."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        k = (r.get("Kernel_Name", "")[:70], r.get("Counter_Name", ""))
        agg[k][0] += 1
        agg[k][1] += float(r.get("Counter_Value", 0) or 0)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("kernel,counter,launches,total,mean_per_launch")
for (k, c), (n, tot) in rows[:120]:
    print('"%s",%s,%d,%.6g,%.6g' % (k, c, n, tot, tot / max(n, 1)))
