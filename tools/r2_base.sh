set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_base; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/dec -- python3 $R/tools/decode_trace.py 15395 24 > $O/dec.txt 2>&1
F=$(find $O/dec -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_summary.py $F 177 > $O/dec_summary.txt 2>&1
cut -c1-1500 $O/bench.json; cat $O/dec_summary.txt; tail -2 $O/dec.txt
find $O/dec -name "*kernel_trace.csv" -size +20M -delete
