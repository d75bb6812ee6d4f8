#!/bin/bash
# ON THE GPU BOX: pipelined ViT attention kernel (COGS_ATTN_VIT=2, default) vs the unpipelined one (=1) and the general
# kernel (=0); correctness tests first
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/attn_ab; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -3 $O/tests.log
for v in 0 1 2; do
  ATTN_PRE=1 COGS_ATTN_VIT=$v timeout -k 10 200 python tools/attn_bench.py > $O/v$v.txt 2>&1
  echo "--- COGS_ATTN_VIT=$v"; grep "vit hd72" $O/v$v.txt
done
