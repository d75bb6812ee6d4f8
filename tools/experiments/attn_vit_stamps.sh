R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/attn_stamps; mkdir -p $O; cd $R
for f in PHASE PIPE; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fno-honor-nans --offload-arch=gfx950 -DCOGS_${f}_STAMPS -I cogstream_amd/csrc tools/micro/attn_vit_micro.cpp -o $O/m_$f 2> $O/build_$f.log || tail -3 $O/build_$f.log; done
for rep in 1 2; do for f in PHASE PIPE; do timeout -k 10 60 $O/m_$f 64 924 1; done; done
