#!/bin/bash
# GPU session r03e: phase stamps of the association kernel (diagnostic build) in grid mode, dense and sparse.
set -o pipefail
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
for n in 150 40 6; do
  echo "== diag, $n frames, GS_GRID_MODE=2"; date
  GS_GRID_MODE=2 timeout -k 10 200 python tools/knn_diag_long.py $n > $O/diag_m2_$n.txt 2>&1; echo "rc=$?"; head -16 $O/diag_m2_$n.txt
done
echo "== diag, 6 frames, default (chunk boxes)"; date
timeout -k 10 200 python tools/knn_diag_long.py 6 > $O/diag_m1_6.txt 2>&1; head -16 $O/diag_m1_6.txt
date
