#!/bin/bash
# GPU session r05b: icp_prepare_k's block 0 with its small inputs requested at once (pix_scan_k / icp_step_k batching reverted)
set -o pipefail
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "grid or knn or icp or c3_64 or dense_regime or localize or straggler or fixture or config1" > $O/tests_sel.log 2>&1; echo "selected tests rc=$?"; tail -2 $O/tests_sel.log
GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; grep frames/s $O/pf200_prof.txt
date
