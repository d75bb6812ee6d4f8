#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of tools/encode_steps.py "$@" and the timeline summary of its last step -> gpurun_out/$OUT
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${OUT:-timeline}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/raw
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 $R/tools/encode_steps.py "$@" > $OUT/stdout.log 2> $OUT/stderr.log
T=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python3 $R/tools/step_timeline.py $T --list 60 --ends 24 > $OUT/timeline.txt
rm -rf $OUT/raw
cat $OUT/stdout.log $OUT/timeline.txt
