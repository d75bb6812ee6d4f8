#!/bin/bash
# Runs ON THE GPU BOX: kernel stats of one rank's encoder step at several clip sizes, ping-pong against the 256x128 ring
# kernel (the calibration behind cogs_k_gemm's few-tile choice)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
for spec in "16 10 20" "32 10 20" "64 10 20" "128 10 20" "8 22 42" "16 22 42" "32 22 42"; do
  tag=$(echo $spec | tr ' ' '_')
  $R/tools/prof_cmd.sh r3a/cal/pp_$tag $R/tools/shard_step.py $spec 5 || exit 1
  $R/tools/prof_cmd.sh r3a/cal/ring_$tag $R/tools/shard_step.py $spec 5 --debug gemm_pingpong=0 || exit 1
done
