#!/bin/bash
# Runs ON THE GPU BOX: launch time of the N = 1152 / 3456 ViT GEMM shapes over the tile-walk group size, tall tiles on / off
# (python tools/gemm_one.py prints the median of 6 launches), then the fabric traffic of the out-projection shape.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for shape in "59136 1152 1152 stat" "59136 1152 4352 stat" "59136 3456 1152 plain" "29568 1152 1152 stat" "29568 1152 4352 stat" "7392 1152 1152 stat" "7392 1152 4352 stat"; do
  for dbg in "gemm_tall=0" "gemm_tall=1" "gemm_group_m=3" "gemm_group_m=6" "gemm_group_m=9" "gemm_group_m=12" "gemm_group_m=6,gemm_split=0" "gemm_tall=0,gemm_split=0"; do
    timeout -k 10 120 python3 tools/gemm_one.py $shape --debug $dbg 2>/dev/null | grep "^M="
  done
done
