#!/usr/bin/env python3
"""matrix-pipe busy share per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
    python tools/mfma_busy.py <dir>
GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = elapsed cycles); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs.
mfma_busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-llm --no-cpu --no-cfg3")
print("# kernel | dispatches | elapsed cycles (GUI_ACTIVE/8) | MFMA busy cycles (sum over 1024 SIMDs) | mfma_busy")
for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 1])[0]):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    b, nb = cs["SQ_VALU_MFMA_BUSY_CYCLES"]; g, ng = cs["GRBM_GUI_ACTIVE"]
    if b <= 0:
        continue
    el = g / ng / 8
    print(f"{k:62s} | {nb:4d} | {el:11.0f} | {b / nb:14.0f} | {100 * (b / nb) / (el * 1024):5.1f} %")
