#!/bin/bash
# GPU session r03t: hook experiment reverted; sequence node on a batch; whole suite.
set -o pipefail
O=gpurun_out/r03t; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -6 $O/gpu_tests.log
GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
date
