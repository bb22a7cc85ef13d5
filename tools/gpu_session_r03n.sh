#!/bin/bash
# GPU session r03n: sequence backward accumulating in place; gradient tests; kernel profile of forward + backward.
set -o pipefail
O=gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gradient or backward or config3 or config5 or sequence_node or fixture_full" > $O/tests_grad.log 2>&1; echo "rc=$?"; tail -3 $O/tests_grad.log
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 gradicp 2>&1 | tail -3
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 icp 2>&1 | tail -1
echo "== fwd+bwd under rocprofv3"; date
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fb -- python3 tools/fwd_bwd_c3.py 200 gradicp > $O/fb_prof.txt 2>&1; tail -1 $O/fb_prof.txt
find $O -name "*kernel_trace.csv" -size +30M -delete
date
