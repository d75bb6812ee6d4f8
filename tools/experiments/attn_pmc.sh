#!/bin/bash
# ON THE GPU BOX: SQ counters of the prompt attention (tools/attn_prefill_ab.py, default kernel only), one --pmc pass per
# counter group (kernel-trace only). usage: attn_pmc.sh [extra args of attn_prefill_ab.py]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/attn_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1 || true
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_CVT SQ_IFETCH"; do
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 $R/tools/attn_prefill_ab.py "$@" > $O/g$i.out 2> $O/g$i.err || echo "group $i failed: $grp"
  i=$((i+1))
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k:
            continue
        acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} mean per launch {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
