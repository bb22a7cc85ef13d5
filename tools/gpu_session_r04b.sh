#!/bin/bash
# GPU session r04b: full GPU tests + 200-frame profile of the restructured association prologue
set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 gradicp 2>&1 | tail -1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; grep frames/s $O/pf200_prof.txt
python tools/trace_assoc.py $O/prof_pf200/*/*_kernel_trace.csv > $O/assoc_by_position.txt 2>&1; tail -8 $O/assoc_by_position.txt
date
