#!/bin/bash
# GPU session r03p: reverse pass with its O(1) steps folded into the wide kernels (two launches per gradLM iteration).
set -o pipefail
O=gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_reference_scenarios.py -q -m gpu -k "grad or backward or config3 or config5 or sequence_node or fixture_full or config1 or icp_grads or provider" > $O/tests_grad.log 2>&1; echo "rc=$?"; tail -5 $O/tests_grad.log
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 gradicp 2>&1 | tail -2
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 icp 2>&1 | tail -1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fb -- python3 tools/fwd_bwd_c3.py 200 gradicp > $O/fb_prof.txt 2>&1; tail -1 $O/fb_prof.txt
find $O -name "*kernel_trace.csv" -size +30M -delete
date
