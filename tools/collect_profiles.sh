#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 summaries behind bench.py's numbers.
#   1. kernel-trace + stats of the encoder-only bench command (what roofline.avg_launch_ms is checked against)
#   2. kernel-trace + stats of the default bench command
#   3./4. FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), kernel-trace only
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc -- python3 $R/bench.py --steps 3 --warmup 1 --no-llm --no-cpu > $OUT/enc.json 2> $OUT/enc.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -- python3 $R/bench.py --steps 3 --warmup 1 > $OUT/full.json 2> $OUT/full.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-llm --no-cpu > /dev/null 2> $OUT/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-llm --no-cpu > /dev/null 2> $OUT/write.err
# keep the merge small: drop the raw traces of the two stats runs except the stats files
find $OUT/full $OUT/enc -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT/*/* | head -30
cat $OUT/enc.json | cut -c1-900
