#!/bin/bash
# GPU session r03z: does HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory) shorten the dependent launches?
set -o pipefail
O=gpurun_out/r03z; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
done
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 python tools/knn_diag_long.py 150 > $O/diag_150_devkernarg.txt 2>&1; grep "block(s)\|span\|block total\|prologue" $O/diag_150_devkernarg.txt
date
