#!/bin/bash
# GPU session r05g: last checks on HEAD -- a wider fuzz of the grid search against brute force (the association kernel's
# prologue was rebuilt this session), smoke, the full GPU suite
set -o pipefail
O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python tools/fuzz_grid.py 1000 240 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
date
