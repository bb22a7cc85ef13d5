#!/bin/bash
# GPU session r03b: fused correspondence chain (flag fix) + geometric-proof grid search: tests, then kernel profiles.
set -o pipefail
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
echo "== grid / tile / straggler / sequence-equality tests"; date
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "grid_search or tile_points or straggler or streamed_arena or reproducible or fixture_full_slam or c3_64" > $O/tests_a.log 2>&1; echo "rc=$?"
grep -E "passed|failed|pose rel err|overflow|gradients:" $O/tests_a.log | tail -30
echo "== pf200 icp under rocprofv3"; date
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; echo "rc=$?"; tail -1 $O/pf200_prof.txt
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp > $O/pf200_clean.txt 2>&1; tail -1 $O/pf200_clean.txt
timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp > $O/pf200_clean_gradicp.txt 2>&1; tail -1 $O/pf200_clean_gradicp.txt
echo "== all gpu tests"; date
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -8 $O/gpu_tests.log
echo "== bench (short)"; date
GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err; echo "bench rc=$?"; cut -c1-400 $O/bench_short.json
find $O -name "*kernel_trace.csv" -size +30M -delete
date
