#!/bin/bash
# GPU session r03k: grid search at every density as the policy (density / small-tile plumbing removed, camera copied into the
# workspace); what a kernel boundary costs (launch_gap micro).
set -o pipefail
O=gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
echo "== launch_gap"; date
timeout -k 10 120 ./tools/micro/launch_gap 10 300 > $O/launch_gap.txt 2>&1; cat $O/launch_gap.txt
echo "== all gpu tests"; date
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -8 $O/gpu_tests.log
echo "== graph mode determinism"; 
GS_GRAPH=1 timeout -k 10 200 python tools/profile_pointfusion.py 60 icp 2>&1 | tail -1
timeout -k 10 200 python tools/profile_pointfusion.py 60 icp 2>&1 | tail -1
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -2
timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp 2>&1 | tail -1
GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err; cut -c1-300 $O/bench_short.json
date
