#!/bin/bash
# GPU session r03j: clean-run launch gaps -- SDMA copies / per-frame count read-backs?
set -o pipefail
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
export GS_GRID_MODE=2
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -2 | tr '\n' ' '; echo; }
run GS_X=0
run HSA_ENABLE_SDMA=0
run GS_NO_READBACK=1
run GS_NO_READBACK=1 HSA_ENABLE_SDMA=0
run GS_X=0
date
