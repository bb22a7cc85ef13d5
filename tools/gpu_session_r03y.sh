#!/bin/bash
# GPU session r03y: association block time by co-residency (CU stamp in the proven path of the diagnostic build),
# and the same with the partial rows summed by a launch of their own (GS_DIAG_PRESUM=1: one row read per block)
set -o pipefail
O=gpurun_out/r03y; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python tools/knn_diag_long.py 150 > $O/diag_150.txt 2>&1; echo "rc=$?"; grep "block(s)\|span\|block total\|prologue" $O/diag_150.txt
GS_DIAG_PRESUM=1 timeout -k 10 200 python tools/knn_diag_long.py 150 > $O/diag_150_presum.txt 2>&1; echo "rc=$?"; grep "block(s)\|span\|block total\|prologue" $O/diag_150_presum.txt
date
