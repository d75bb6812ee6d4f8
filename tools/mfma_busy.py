#!/usr/bin/env python3
"""matrix-pipe busy share per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
    python tools/mfma_busy.py <dir>
GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = elapsed cycles); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs.
mfma_busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
For the three ViT GEMM kernels of the cfg2 encoder-only command (M = 59 136) the last column is the USEFUL share: the
matrix-pipe cycles the algorithmic FLOPs need (2 M N K / 1024: a v_mfma_f32_16x16x32_bf16 is 16 384 FLOP in 16 cycles)
over the same SIMD cycles -- busy minus useful = MFMAs spent on tile padding."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-llm --no-cpu --no-cfg3")
print("# kernel | dispatches | elapsed cycles (GUI_ACTIVE/8) | MFMA busy cycles (sum over 1024 SIMDs) | mfma_busy | algorithmic MFMA cycles, useful share (cfg2)")
M = 59136
ALGO = {"gemm_tn_pp64_kernel<2057>": 2.0 * M * 4352 * 1152 / 1024,
        "gemm_tn_pp64_kernel<6149>": 2.0 * M * 3456 * 1152 / 1024}
for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 1])[0]):
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in cs or "GRBM_GUI_ACTIVE" not in cs:
        continue
    b, nb = cs["SQ_VALU_MFMA_BUSY_CYCLES"]; g, ng = cs["GRBM_GUI_ACTIVE"]
    if b <= 0:
        continue
    el = g / ng / 8
    algo = next((v for n, v in ALGO.items() if k.startswith(n)), None)
    extra = f" | {algo:14.0f} | {100 * algo / (el * 1024):5.1f} % (padding {100 * (b / nb / algo - 1):+5.1f} %)" if algo else ""
    print(f"{k:62s} | {nb:4d} | {el:11.0f} | {b / nb:14.0f} | {100 * (b / nb) / (el * 1024):5.1f} %{extra}")

# the N = 1152 GEMMs (EPI 1027: out-proj of all 27 layers, fc2 of the first 26) may be split between the ping-pong and the ring
# body (round-aligned split of a one-stream launch): compare their TOTAL with the algorithmic count of a pass
tot1027 = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"][0] for k, cs in acc.items() if "<1027>" in k or "1027>" in k)
passes = acc.get("gemm_tn_pp64_kernel<2057>(GemmArgs)", {}).get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0])[1] / 27.0
if passes > 0 and tot1027 > 0:
    algo = passes * (27 * 2.0 * M * 1152 * 1152 + 26 * 2.0 * M * 1152 * 4352) / 1024
    print(f"# EPI 1027 kernels (both bodies) over {passes:.0f} passes: MFMA busy cycles {tot1027:.0f}, algorithmic {algo:.0f} "
          f"(padding {100 * (tot1027 / algo - 1):+.2f} %)")
