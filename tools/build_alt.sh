#!/bin/bash
# Build cogstream_amd/libcogs_hip_alt.so: the library with ONE source (first argument: gemm, gemv, ...) compiled under
# extra flags (the remaining arguments), for in-process A/B runs (tools/gemm_ab_lib.py, tools/gemv_bench.py with
# COGS_ALT_LIB=1). Everything else comes from the default build's objects.
set -eu
cd "$(dirname "$0")/.."
python -m cogstream_amd.build > /dev/null
B=cogstream_amd/csrc/build
SRC=$1; shift
EXTRA=""; [ "$SRC" = attn_vit ] && EXTRA="-fno-honor-nans"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $EXTRA "$@" -c cogstream_amd/csrc/$SRC.hip -o $B/${SRC}_${ALT_NAME:-alt}.o
objs=""
for f in cogstream_amd/csrc/*.hip; do n=$(basename $f .hip); if [ $n = $SRC ]; then objs="$objs $B/${SRC}_${ALT_NAME:-alt}.o"; else objs="$objs $B/$n.o"; fi; done
OUT=cogstream_amd/libcogs_hip_${ALT_NAME:-alt}.so      # ALT_NAME=x: several alternatives side by side (tools/gemm_ab_lib.py x y ...)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT $objs
echo $OUT
