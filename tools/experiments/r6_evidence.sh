#!/bin/bash
# Runs ON THE GPU BOX: the LLM-side evidence of round 6 on the current binary -> gpurun_out/r6_evidence/
#   prefill anatomy (kernel trace), prefill MFMA-busy (PMC pass), decode step anatomy, decode traffic (PMC pass), MFMA-shape clocks
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_evidence; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tr && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/decode_trace.py 15395 4 > $O/prefill_trace.out 2> $O/prefill_trace.err
python3 $R/tools/prefill_anatomy.py $(find $O/tr -name "*kernel_trace.csv" | head -1) > $O/prefill_anatomy.txt
rm -rf $O/tr
rm -rf $O/pmc && timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python3 $R/tools/decode_trace.py 15395 4 > /dev/null 2> $O/prefill_pmc.err
python3 $R/tools/mfma_busy.py $O/pmc > $O/prefill_pmc_mfma_busy.txt
rm -rf $O/pmc
rm -rf $O/tr && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/decode_trace.py 15395 12 > $O/decode_trace.out 2> $O/decode_trace.err
python3 $R/tools/decode_step_anatomy.py $(find $O/tr -name "*kernel_trace.csv" | head -1) > $O/decode_step_anatomy.txt
rm -rf $O/tr
rm -rf $O/pmc && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python3 $R/tools/decode_trace.py 15395 12 > /dev/null 2> $O/decode_pmc.err
python3 $R/tools/decode_traffic.py $(find $O/pmc -name "*counter_collection.csv" | head -1) 15400 > $O/decode_traffic.json
rm -rf $O/pmc
timeout -k 10 120 $R/tools/micro/mfma_shape_dvfs > $O/mfma_shape_dvfs.txt 2>&1
tail -n 20 $O/prefill_anatomy.txt; cat $O/prefill_pmc_mfma_busy.txt | head -20; head -20 $O/decode_step_anatomy.txt; cat $O/decode_traffic.json; cat $O/mfma_shape_dvfs.txt
