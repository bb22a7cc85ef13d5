#!/bin/bash
# GPU session r03l: bisect the clean run's gaps; stamps with the early row loads.
set -o pipefail
O=gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python tools/gap_bisect.py 200 2>&1 | tail -9
timeout -k 10 200 python tools/knn_diag_long.py 150 > $O/diag_150.txt 2>&1; sed -n 3,12p $O/diag_150.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "grid_search or tile_points or straggler or dense_regime or reproducible" 2>&1 | tail -2
GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
date
