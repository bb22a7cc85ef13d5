#!/bin/bash
# GPU session r04a: association kernel with every kernel-start request in one batch (no loop-constant trip in front)
set -o pipefail
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "grid or knn or icp or c3_64 or dense_regime or localize or straggler" > $O/tests_sel.log 2>&1; echo "selected tests rc=$?"; tail -3 $O/tests_sel.log
timeout -k 10 200 python tools/knn_diag_long.py 150 > $O/diag_150.txt 2>&1; grep "block(s)\|span\|block total\|prologue" $O/diag_150.txt
GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp 2>&1 | tail -1
date
