#!/bin/bash
# GPU session r03h: what makes the clean run slower than the profiled one?  Runtime knobs on the same 200-frame run;
# phase stamps of the staging path.
set -o pipefail
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1; }
run GS_X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_DIRECT_DISPATCH=0
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000000
run GS_GRAPH=1
run HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0 GPU_MAX_HW_QUEUES=2
run GS_X=0
echo "== under rocprofv3 (kernel trace only)"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 tools/profile_pointfusion.py 200 icp 2>&1 | grep frames/s
echo "== staging stamps"
GS_GRID_MODE=2 timeout -k 10 200 python tools/knn_diag_long.py 150 > $O/diag_m2_150.txt 2>&1; sed -n 3,14p $O/diag_m2_150.txt
nproc; lscpu | grep -E "Model name|MHz" | head -4
date
