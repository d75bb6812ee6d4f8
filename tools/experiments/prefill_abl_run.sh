#!/bin/bash
# ON THE GPU BOX: time the prompt attention with each ablation library (tools/experiments/prefill_abl.sh), the shipped one first and last
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== shipped"; timeout -k 10 100 python tools/attn_prefill_ab.py 15395 2>&1 | grep variant
for f in cogstream_amd/abl/libcogs_*.so; do
  echo "== $(basename $f)"; COGS_LIB_PATH=$R/$f timeout -k 10 100 python tools/attn_prefill_ab.py 15395 2>&1 | grep variant
done
echo "== shipped"; timeout -k 10 100 python tools/attn_prefill_ab.py 15395 2>&1 | grep variant
