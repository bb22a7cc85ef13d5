#!/bin/bash
# GPU session r04d: the frame's ds-grid source cloud on the launches of the map's projection (two launches less per step)
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp 2>&1 | tail -1
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 gradicp 2>&1 | tail -1
date
