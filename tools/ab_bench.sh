#!/bin/bash
# ON THE GPU BOX: encoder-only bench, A/B by environment variable (interleaved twice)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2; do
  for v in "$@"; do
    echo "== $v"; env $v python bench.py --steps 10 --warmup 3 --no-llm --no-cpu --no-cfg3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['breakdown_ms'], d.get('attention_tflops'))"
  done
done
