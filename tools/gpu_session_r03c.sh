#!/bin/bash
# GPU session r03c: the grid search at every density (GS_GRID_MODE=2) against the default policy, on c2 and on 200 frames;
# the new dense-regime step test.
set -o pipefail
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
echo "== dense-regime step vs oracle"; date
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "dense_regime_step" > $O/tests_dense.log 2>&1; echo "rc=$?"
grep -E "passed|failed|pose rel err|merged rows|Error" $O/tests_dense.log | tail -20
for mode in 1 2; do
  echo "== bench short, GS_GRID_MODE=$mode"; date
  GS_GRID_MODE=$mode GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_short_m$mode.json 2> $O/bench_short_m$mode.err; echo "rc=$?"; cut -c1-330 $O/bench_short_m$mode.json
  GS_GRID_MODE=$mode timeout -k 10 200 python tools/profile_pointfusion.py 200 icp > $O/pf200_m$mode.txt 2>&1; tail -1 $O/pf200_m$mode.txt
  GS_GRID_MODE=$mode timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp > $O/pf200g_m$mode.txt 2>&1; tail -1 $O/pf200g_m$mode.txt
done
echo "== pf200 icp mode 2 under rocprofv3"; date
GS_GRID_MODE=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200_m2 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof_m2.txt 2>&1; echo "rc=$?"; grep frames/s $O/pf200_prof_m2.txt
date
