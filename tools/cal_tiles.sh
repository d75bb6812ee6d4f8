set -u
for spec in "16 10 20" "32 10 20" "64 10 20" "128 10 20" "8 22 42" "16 22 42" "32 22 42"; do
  tag=$(echo $spec | tr ' ' '_')
  tools/prof_cmd.sh r3a/cal/pp_$tag $GRAFT_REPO_ROOT/tools/shard_step.py $spec 5 || exit 1
  COGS_GEMM_NOPP=1 tools/prof_cmd.sh r3a/cal/ring_$tag $GRAFT_REPO_ROOT/tools/shard_step.py $spec 5 || exit 1
done
