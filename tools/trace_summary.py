#!/usr/bin/env python3
"""summarise a rocprofv3 kernel_trace.csv: per-kernel totals + the last N kernels (one decode step)"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 260
GRID = "Grid_Size_X" if rows and "Grid_Size_X" in rows[0] else ("Grid_Size" if rows and "Grid_Size" in rows[0] else None)
def short(n, r=None):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if r is not None and GRID and n.startswith("gemv_kernel"):
        return (n[:30] + " grid=" + r[GRID])[:44]
    return n[:44]
tot = collections.Counter(); cnt = collections.Counter()
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[short(r["Kernel_Name"])] += d; cnt[short(r["Kernel_Name"])] += 1
print("== all kernels")
for k, v in tot.most_common(14):
    print(f"{k:46s} n={cnt[k]:6d} total={v/1e3:9.2f} ms avg={v/cnt[k]:8.1f} us")
last = rows[-n_last:]
t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
lt = collections.Counter(); lc = collections.Counter()
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    lt[short(r["Kernel_Name"], r)] += d; lc[short(r["Kernel_Name"], r)] += 1
print(f"== last {n_last} kernels: span {(t1-t0)/1e6:.3f} ms, busy {sum(lt.values())/1e3:.3f} ms")
for k, v in lt.most_common(16):
    print(f"{k:46s} n={lc[k]:4d} total={v:8.1f} us avg={v/lc[k]:7.1f} us")
