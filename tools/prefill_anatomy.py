#!/usr/bin/env python3
"""The last prompt pass out of a rocprofv3 kernel trace of tools/decode_trace.py: the kernels between the last two lm_head
GEMVs, per kernel name: count, average duration, total."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "gemv_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == 152064 // 16 * 256]
a, b = heads[-2], heads[-1]
step = rows[a + 1:b + 1]
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
tot, cnt = collections.Counter(), collections.Counter()
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    key = f"{n[:60]}"
    tot[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[key] += 1
print(f"last prefill: {len(step)} kernels, span {span/1e3:.2f} ms, busy {sum(tot.values())/1e3:.2f} ms")
for k, v in tot.most_common(16):
    print(f"{k:62s} n={cnt[k]:3d} avg={v / cnt[k]:9.2f} us total={v/1e3:8.2f} ms")
