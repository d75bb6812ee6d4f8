set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/attn_len; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc -O3 -std=c++17 -fno-honor-nans --offload-arch=gfx950 -I cogstream_amd/csrc tools/micro/attn_vit_micro.cpp -o $O/m 2> $O/build.log || { tail -5 $O/build.log; exit 1; }
/opt/rocm/bin/hipcc -O3 -std=c++17 -fno-honor-nans --offload-arch=gfx950 -DCOGS_LIFE_STAMPS -I cogstream_amd/csrc tools/micro/attn_vit_micro.cpp -o $O/ml 2> $O/buildl.log || { tail -5 $O/buildl.log; exit 1; }
for rep in 1 2 3; do for v in 0 1; do echo "== attn_vit_len=$v"; timeout -k 10 60 $O/m 64 924 1 $v; timeout -k 10 60 $O/ml 64 924 1 $v | tail -1; done; done
