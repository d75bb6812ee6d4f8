#!/bin/bash
# ON THE GPU BOX: the encoder step with the shipped library and with cogstream_amd/libcogs_hip_alt.so (tools/build_alt.sh),
# alternating processes A B A B on one box. usage: lib_abab.sh [args of tools/encoder_ab.py]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  echo "== shipped"; timeout -k 10 200 python tools/encoder_ab.py "$@" 2>&1 | grep "step median"
  echo "== alt"; COGS_LIB_PATH=$R/cogstream_amd/libcogs_hip_alt.so timeout -k 10 200 python tools/encoder_ab.py "$@" 2>&1 | grep "step median"
done
