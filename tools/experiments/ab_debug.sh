#!/bin/bash
# Runs ON THE GPU BOX: in-run A/B of one library debug switch (csrc/debug.h) on the encoder bench, interleaved pairs on
# the same box.   ab_debug.sh name=value[,name=value] [pairs] [extra bench flags]   (A = shipped defaults, B = with the switches)
set -u
kv=$1; n=${2:-3}; shift; shift || true
for i in $(seq 1 $n); do
  for side in A B; do
    if [ $side = B ]; then dbg="--debug $kv"; else dbg=""; fi
    echo -n "$side $i: "
    python bench.py --no-cpu --no-pipeline --no-llm $dbg "$@" 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print(d['value'], d['ms_per_step'], 'gemm', d['breakdown_ms']['gemm'], 'attn', d['breakdown_ms']['attention'], 'frac', d['roofline']['frac'], 'cfg3', d['cfg3']['ms_per_step'], 'shard8', d['shard8']['ms_per_step'], d['cfg3']['shard8']['ms_per_step'])"
  done
done
