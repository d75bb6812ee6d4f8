#!/bin/bash
# Runs ON THE GPU BOX: in-run A/B of one environment switch on the encoder bench, interleaved pairs on the same box.
#   ab_env.sh VAR=VALUE [pairs]      (A = without the variable, B = with it)
set -u
kv=$1; n=${2:-3}
for i in $(seq 1 $n); do
  for side in A B; do
    if [ $side = B ]; then export "$kv"; else unset "${kv%%=*}"; fi
    echo -n "$side $i: "
    python bench.py --no-cpu --no-pipeline --no-llm 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print(d['value'], d['ms_per_step'], 'gemm', d['breakdown_ms']['gemm'], 'attn', d['breakdown_ms']['attention'], 'frac', d['roofline']['frac'], 'cfg3', d['cfg3']['ms_per_step'], 'shard8', d['shard8']['ms_per_step'], d['cfg3']['shard8']['ms_per_step'])"
  done
done
