#!/bin/bash
# GPU session r03v: the folded step's partial rows summed by four waves without block barriers (LDS flags), the other twelve
# staging from kernel entry; batch sequence node with arena growth.
set -o pipefail
O=gpurun_out/r03v; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "grid_search or tile_points or straggler or dense_regime or reproducible or streamed_arena or config5 or full_size or sequence_node or config4" > $O/tests_a.log 2>&1; echo "rc=$?"; tail -4 $O/tests_a.log
timeout -k 10 200 python tools/knn_diag_long.py 150 > $O/diag_150.txt 2>&1; sed -n 3,12p $O/diag_150.txt
timeout -k 10 200 python tools/knn_diag_long.py 6 > $O/diag_6.txt 2>&1; sed -n 3,12p $O/diag_6.txt
GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp 2>&1 | tail -1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; grep frames/s $O/pf200_prof.txt
date
