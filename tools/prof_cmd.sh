#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of one python command. Usage: prof_cmd.sh <outdir-under-gpurun_out> <script> [args...]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 "$@" > $OUT/stdout.log 2> $OUT/stderr.log
rc=$?
S=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp $S $OUT/kernel_stats.csv
rm -rf $OUT/raw
cat $OUT/stdout.log
exit $rc
