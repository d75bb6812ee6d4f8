#!/usr/bin/env python3
"""Timeline of the LAST encoder step in a rocprofv3 kernel_trace.csv of tools/encode_steps.py (steps are separated by
pauses > 5 ms): how many kernels run at a time, how long each class runs, where nothing runs.

    python tools/step_timeline.py <kernel_trace.csv> [--list N] [--ends N]"""
import csv
import re
import sys
from collections import Counter

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into bursts at pauses > 5 ms, keep the last burst with > 100 kernels
bursts, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in cur[-8:]) > 5e6:
        bursts.append(cur)
        cur = []
    cur.append(b)
bursts.append(cur)
step = [b for b in bursts if len(b) > 100][-1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<[^>(]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:40]


t0 = min(int(r["Start_Timestamp"]) for r in step)
t1 = max(int(r["End_Timestamp"]) for r in step)
ev = []
for r in step:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, hist = 0, t0, Counter()
for t, d in ev:
    hist[depth] += t - last
    depth += d
    last = t
span = (t1 - t0) / 1e6
print(f"step: {len(step)} kernels, span {span:.3f} ms")
for k in sorted(hist):
    print(f"  {k} kernels resident: {hist[k] / 1e6:7.3f} ms ({100 * hist[k] / (t1 - t0):5.1f} %)")
tot, cnt = Counter(), Counter()
for r in step:
    n = short(r["Kernel_Name"])
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[n] += 1
print("  kernel-time by class (overlapping kernels each count their own span):")
for n, v in tot.most_common(12):
    print(f"    {n:44s} n={cnt[n]:4d} total {v / 1e6:7.3f} ms avg {v / cnt[n] / 1e3:7.1f} us")
def show(rs, title):
    print(f"  {title} (start us, duration us, stream, grid, name):")
    for r in rs:
        print(f"    {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} "
              f"s{r['Stream_Id']} q{r['Queue_Id']} g{int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d} {short(r['Kernel_Name'])}")


if "--list" in sys.argv:
    nlist = int(sys.argv[sys.argv.index("--list") + 1])
    skip = len(step) // 2
    show(step[skip:skip + nlist], f"kernels {skip}..{skip + nlist} of the step")
if "--ends" in sys.argv:
    n = int(sys.argv[sys.argv.index("--ends") + 1])
    show(step[:n], f"first {n} kernels")
    show(step[-n:], f"last {n} kernels")
