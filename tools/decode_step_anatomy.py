#!/usr/bin/env python3
"""One decode step out of a rocprofv3 kernel trace of tools/decode_trace.py: the kernels between two consecutive lm_head
GEMVs (grid 152064 / 16 workgroups), per kernel name and grid: count, average duration, total."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "gemv_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == 152064 // 16 * 256]
a, b = heads[len(heads) // 2], heads[len(heads) // 2 + 1]
step = rows[a + 1:b + 1]
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
tot, cnt = collections.Counter(), collections.Counter()
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    key = f"{n[:44]} grid={int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))}"
    tot[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[key] += 1
print(f"decode step: {len(step)} kernels, span {span:.1f} us, busy {sum(tot.values()):.1f} us")
for k, v in tot.most_common(20):
    print(f"{k:64s} n={cnt[k]:3d} avg={v / cnt[k]:7.2f} us total={v:8.1f} us")
