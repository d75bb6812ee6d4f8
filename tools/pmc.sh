#!/bin/bash
# Runs ON THE GPU BOX: tools/pmc.sh <python script> <kernel-name filter> <counter group> [<counter group> ...]
# Each group ("A B C") is one rocprofv3 --pmc pass (kernel-trace only, as gpurun requires).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
S=$1; F=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  O=$R/gpurun_out/pmc_$i; rm -rf $O; mkdir -p $O
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O -- python3 $R/$S > $O/stdout.txt 2> $O/stderr.txt
  python3 $R/tools/pmc_summary.py $O "$F"
  i=$((i+1))
done
