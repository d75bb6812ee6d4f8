#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel: python tools/pmc_summary.py <dir> [name-substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if flt and flt not in k:
                continue
            k = k[:70]
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print(f"   {c:32s} avg {s / n:16.1f}   over {n} dispatches")
