#!/bin/bash
# ON THE GPU BOX: phase stamps of the ViT attention's main loop (tools/micro/attn_vit_micro built with -DCOGS_PHASE_STAMPS)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/attn_vit_phases; mkdir -p $O
cd $R
/opt/rocm/bin/hipcc -O3 -std=c++17 -fno-honor-nans --offload-arch=gfx950 -DCOGS_PHASE_STAMPS -I cogstream_amd/csrc tools/micro/attn_vit_micro.cpp -o $O/m 2> $O/build.log || { tail -5 $O/build.log; exit 1; }
/opt/rocm/bin/hipcc -O3 -std=c++17 -fno-honor-nans --offload-arch=gfx950 -I cogstream_amd/csrc tools/micro/attn_vit_micro.cpp -o $O/m0 2>> $O/build.log
for rep in 1 2; do timeout -k 10 60 $O/m0 64 924 1; timeout -k 10 60 $O/m 64 924 1; timeout -k 10 60 $O/m 16 3696 1; done
