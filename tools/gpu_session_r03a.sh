#!/bin/bash
# GPU session r03a: the round's first look at HEAD -- new reference-golden tests, launch_chain, bench + profiles of HEAD.
# gpurun --timeout 1200 -- 'bash tools/gpu_session_r03a.sh'
set -o pipefail
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
echo "== new tests"; date
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "c3_64_frames or fixture_full_slam or straggler_search" > $O/new_tests.log 2>&1; echo "new tests rc=$?" | tee -a $O/new_tests.log
tail -5 $O/new_tests.log
echo "== all gpu tests"; date
timeout -k 10 500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/gpu_tests.log
tail -15 $O/gpu_tests.log
echo "== launch_chain"; date
timeout -k 10 60 ./tools/micro/launch_chain 300 12 220 > $O/launch_chain.txt 2>&1; echo "rc=$?"; cat $O/launch_chain.txt
timeout -k 10 60 ./tools/micro/launch_chain 300 3 220 > $O/launch_chain_3us.txt 2>&1; cat $O/launch_chain_3us.txt
echo "== bench (clean)"; date
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
echo "== bench under rocprofv3"; date
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err; echo "rc=$?"
echo "== pointfusion 200 icp under rocprofv3"; date
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; echo "rc=$?"; tail -2 $O/pf200_prof.txt
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp > $O/pf200_clean.txt 2>&1; tail -1 $O/pf200_clean.txt
# keep the merge-back small: stats + the c2 region's trace only
find $O -name "*kernel_trace.csv" -size +40M -delete
ls -la $O $O/prof_bench/* $O/prof_pf200/* 2>/dev/null | head -40
date
