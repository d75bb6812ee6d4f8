#!/bin/bash
# ON THE GPU BOX: SQ counters of the ViT attention through the stand-alone harness (tools/micro/attn_vit_micro), one --pmc
# pass per counter group (kernel-trace only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/attn_vit_pmc; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc -O3 -std=c++17 -fno-honor-nans --offload-arch=gfx950 -I cogstream_amd/csrc tools/micro/attn_vit_micro.cpp -o $O/m 2> $O/build.log || { tail -5 $O/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_CVT SQ_IFETCH"; do
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- $O/m 64 924 1 > $O/g$i.out 2> $O/g$i.err || echo "group $i failed: $grp"
  i=$((i+1))
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} mean per launch {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
