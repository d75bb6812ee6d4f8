#!/bin/bash
# ON THE GPU BOX: SQ instruction-mix counters of one encoder step per kernel (rocprofv3 --pmc passes, kernel-trace only)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/gemm_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-llm --no-cpu --no-cfg3 --emulate-shard 0 --vit-streams 1 --no-session > $O/g$i.out 2> $O/g$i.err || echo "group $i failed"
  i=$((i+1))
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
m = lambda d, c: (sum(d[c]) / len(d[c])) if c in d and d[c] else 0.0
rows = sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0])))
print("# per launch means. busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024); valu/mfma = non-MFMA VALU instructions per MFMA; valu act = (ACTIVE_INST_VALU * 4 - 4 * INSTS_MFMA) / (GUI/8*1024)")
print(f"{'kernel':64s} | launches | cycles | busy % | valu/mfma | lds/mfma | salu/mfma | valu act % | wave resid % | wait_inst % | wait_any % | lds conflict %")
for k, d in rows:
    g = m(d, "GRBM_GUI_ACTIVE") / 8
    mf = m(d, "SQ_INSTS_MFMA")
    if g <= 0 or mf <= 0:
        continue
    simd = g * 1024
    wc = m(d, "SQ_WAVE_CYCLES")
    print(f"{k:64s} | {len(d['GRBM_GUI_ACTIVE']):4d} | {g:9.0f} | {100 * m(d, 'SQ_VALU_MFMA_BUSY_CYCLES') / simd:5.1f} | {(m(d, 'SQ_INSTS_VALU') - mf) / mf:6.2f} | {m(d, 'SQ_INSTS_LDS') / mf:5.2f} | {m(d, 'SQ_INSTS_SALU') / mf:5.2f} | "
          f"{100 * (m(d, 'SQ_ACTIVE_INST_VALU') * 4 - 4 * mf) / simd:5.1f} | {100 * wc * 4 / (g * 256 * 8):5.1f} | {100 * m(d, 'SQ_WAIT_INST_ANY') / max(wc, 1):5.1f} | {100 * m(d, 'SQ_WAIT_ANY') / max(wc, 1):5.1f} | "
          f"{100 * m(d, 'SQ_LDS_BANK_CONFLICT') / max(m(d, 'SQ_LDS_IDX_ACTIVE'), 1):5.1f}")
PY
