#!/bin/bash
# ON THE GPU BOX: new ViT attention kernel vs the general one (COGS_ATTN_VIT=0), correctness tests first
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/attn_ab; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -3 $O/tests.log
ATTN_PRE=1 COGS_ATTN_VIT=0 timeout -k 10 200 python tools/attn_bench.py > $O/old.txt 2>&1
ATTN_PRE=1 COGS_ATTN_WPS=2 timeout -k 10 200 python tools/attn_bench.py > $O/new2.txt 2>&1
ATTN_PRE=1 COGS_ATTN_WPS=3 timeout -k 10 200 python tools/attn_bench.py > $O/new3.txt 2>&1
echo "--- general kernel"; grep "vit hd72" $O/old.txt; echo "--- attn_vit kernel wps 2"; grep "vit hd72" $O/new2.txt; echo "--- attn_vit kernel wps 3"; grep "vit hd72" $O/new3.txt
