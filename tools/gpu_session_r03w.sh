#!/bin/bash
# GPU session r03w: kernel arguments preloaded into SGPRs (-amdgpu-kernarg-preload-count=16), four-wave reduce reverted.
set -o pipefail
O=gpurun_out/r03w; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout -k 10 200 python tools/knn_diag_long.py 150 > $O/diag_150.txt 2>&1; sed -n 3,12p $O/diag_150.txt
GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp 2>&1 | tail -1
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 gradicp 2>&1 | tail -1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; grep frames/s $O/pf200_prof.txt
date
