#!/bin/bash
# ON THE GPU BOX: one decode step at 15.4k context by kernel (rocprofv3 kernel trace of tools/decode_trace.py + tools/decode_step_anatomy.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dec_anatomy; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/decode_trace.py 15395 12 "$@" > $O/run.out 2> $O/run.err
python3 $R/tools/decode_step_anatomy.py $(find $O/tr -name "*kernel_trace.csv" | head -1)
find $O -name "*kernel_trace.csv" -delete
