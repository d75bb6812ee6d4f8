#!/bin/bash
# Runs ON THE GPU BOX: the two-stream encoder step (tools/encoder_ab.py: medians of back-to-back steps) with the default library against
# cogstream_amd/libcogs_hip_$1.so (COGS_LIB_PATH), alternating processes on one box: tools/experiments/lib_step_ab.sh <name> [rounds] [encoder_ab flags]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; ROUNDS=${2:-3}; shift; shift || true
cd $R
for i in $(seq $ROUNDS); do
  echo -n "default: "; timeout -k 10 200 python3 tools/encoder_ab.py --rounds 5 "$@" 2>/dev/null | grep "^default" | cut -c1-150
  echo -n "$N: "; COGS_LIB_PATH=$R/cogstream_amd/libcogs_hip_$N.so timeout -k 10 200 python3 tools/encoder_ab.py --rounds 5 "$@" 2>/dev/null | grep "^default" | cut -c1-150
done
