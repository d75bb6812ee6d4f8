#!/bin/bash
# ON THE GPU BOX: decode-path A/B of two builds of gemv.hip (default vs flags "$@") -- per-shape GEMV times and the
# per-token decode time of the 7B model at 15.4k context; both libraries travel with the snapshot
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2; do
  echo "== default"; timeout -k 10 200 python tools/gemv_bench.py 2>&1 | tail -6
done
timeout -k 10 300 python tools/decode_trace.py 15395 33 2>&1 | tail -1
