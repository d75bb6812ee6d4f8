#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 summaries behind bench.py's numbers. Usage: collect_profiles.sh <tag>
# Since round 4 every clip is encoded as two halves on two streams (overlapping kernels stretch each other's durations in
# a trace); bench.py's own per-kernel pass (`roofline`, `breakdown_ms`) runs on ONE stream, so the per-kernel commands
# here (1, 3, 4, 5) run with --vit-streams 1 as well -- their averages are what `roofline.avg_launch_ms` is checked
# against. Command 2 is the default bench run as the driver starts it.
#   1. kernel-trace + stats of the encoder-only bench command (what roofline.avg_launch_ms is checked against)
#   2. kernel-trace + stats of the default bench command
#   3./4. FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), kernel-trace only
#   5. SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE (matrix-pipe busy share per kernel)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r5}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -- python3 $R/bench.py --steps 3 --warmup 1 > $OUT/full.json 2> $OUT/full.err
set -- --vit-streams 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc -- python3 $R/bench.py --steps 3 --warmup 1 --no-llm --no-cpu --no-cfg3 --emulate-shard 0 "$@" > $OUT/enc.json 2> $OUT/enc.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-llm --no-cpu --no-cfg3 --emulate-shard 0 "$@" > /dev/null 2> $OUT/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 0 --no-llm --no-cpu --no-cfg3 --emulate-shard 0 "$@" > /dev/null 2> $OUT/write.err
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- python3 $R/bench.py --steps 1 --warmup 0 --no-llm --no-cpu --no-cfg3 --emulate-shard 0 "$@" > /dev/null 2> $OUT/mfma.err
F=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 $R/tools/traffic_from_pmc.py $F $W > $OUT/gemm_traffic.json
python3 $R/tools/mfma_busy.py $OUT/mfma > $OUT/pmc_mfma_busy.txt
cp $(find $OUT/enc -name "*kernel_stats.csv" | head -1) $OUT/enc_kernel_stats.csv
cp $(find $OUT/full -name "*kernel_stats.csv" | head -1) $OUT/full_kernel_stats.csv
# keep the merge small: drop the raw traces and counter dumps
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
cat $OUT/pmc_mfma_busy.txt; cat $OUT/gemm_traffic.json; cut -c1-1200 $OUT/enc.json
