#!/bin/bash
# GPU session r04c: reverse pass with the folded steps' inputs requested in one batch (state, record pair, rows -> LDS)
set -o pipefail
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu -x -k "grad or backward or bwd or sequence_node or config1 or config3 or config5 or fixture or tape or autograd" > $O/tests_grad.log 2>&1; echo "gradient tests rc=$?"; tail -3 $O/tests_grad.log
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 gradicp 2>&1 | tail -1
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 icp 2>&1 | tail -1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fb -- python3 tools/fwd_bwd_c3.py 200 gradicp > $O/fb_prof.txt 2>&1; tail -1 $O/fb_prof.txt
date
