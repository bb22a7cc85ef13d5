#!/bin/bash
# GPU session r05d: HEAD (7ef3264: ee311fe's kernels minus the two load batchings that measured slower) -- smoke, the GPU suite, the full bench line
set -o pipefail
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
export GS_COMMIT=7ef3264
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; cut -c1-260 $O/bench.json
date
