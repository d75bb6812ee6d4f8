#!/usr/bin/env python3
"""Fabric-side read bytes of ONE decode step from a `rocprofv3 --pmc FETCH_SIZE --kernel-trace` pass of
tools/decode_trace.py: FETCH_SIZE (KiB, doubled: gfx950 counts a 128-byte request as 64 bytes for wide streaming reads,
MI355X_MICROARCH.md section HBM) summed per kernel name over the dispatches between two consecutive lm_head GEMVs.
    python tools/decode_traffic.py <counter_collection.csv> <context> > decode_traffic.json"""
import collections
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE"]
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
heads = [i for i, r in enumerate(rows) if "gemv_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 152064 // 16 * 256]
a, b = heads[len(heads) // 2], heads[len(heads) // 2 + 1]
step = rows[a + 1:b + 1]
per = collections.Counter()
for r in step:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    per[n] += 2.0 * float(r["Counter_Value"]) * 1024.0
total = sum(per.values())
alg = 2 * (6.526e9 + 545e6) + 57344.0 * ctx
print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/decode_trace.py (one decode step: the "
                            "dispatches between two lm_head GEMVs); FETCH_SIZE doubled per the gfx950 correction",
                  "context": ctx, "dispatches": len(step), "fetch_bytes_per_token": total,
                  "algorithmic_bytes_per_token": alg, "ratio": total / alg,
                  "by_kernel": {k: v for k, v in per.most_common(8)}}, indent=1))
