#!/bin/bash
# Build cogstream_amd/libcogs_hip_alt.so: the library with gemm.hip compiled under extra flags ("$@"), for in-process A/B
# runs of GEMM variants (tools/gemm_ab_lib.py). Everything else comes from the default build's objects.
set -eu
cd "$(dirname "$0")/.."
python -m cogstream_amd.build > /dev/null
B=cogstream_amd/csrc/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c cogstream_amd/csrc/gemm.hip -o $B/gemm_alt.o
objs=""
for f in cogstream_amd/csrc/*.hip; do n=$(basename $f .hip); if [ $n = gemm ]; then objs="$objs $B/gemm_alt.o"; else objs="$objs $B/$n.o"; fi; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o cogstream_amd/libcogs_hip_alt.so $objs
echo cogstream_amd/libcogs_hip_alt.so
