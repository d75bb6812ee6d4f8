#!/bin/bash
# ON THE GPU BOX: A/B builds of the pipelined ViT attention kernel through the stand-alone harness (same box, interleaved)
# usage: attn_abl.sh "<flags A>" "<flags B>" ...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/attn_abl; mkdir -p $O
cd $R
i=0
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fno-honor-nans --offload-arch=gfx950 $v -I cogstream_amd/csrc tools/micro/attn_vit_micro.cpp -o $O/m_$i 2> $O/build_$i.log || { echo "build failed $v"; tail -5 $O/build_$i.log; }
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for v in "$@"; do
    echo "== [$v]"; timeout -k 10 60 $O/m_$i 64 924; timeout -k 10 60 $O/m_$i 16 3696
    i=$((i+1))
  done
done
