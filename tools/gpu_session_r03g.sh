#!/bin/bash
# GPU session r03g: is the legacy default stream what costs ~5 us per launch in clean runs?  Same run on a side stream.
set -o pipefail
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
GS_SIDE_STREAM=1 timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
done
GS_SIDE_STREAM=1 GS_GRID_MODE=2 timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
GS_SIDE_STREAM=1 GS_GRID_MODE=2 GS_GRID_RADIUS=1 timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
GS_SIDE_STREAM=1 timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp 2>&1 | tail -1
timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp 2>&1 | tail -1
echo "== rocprof clean-vs-side"; 
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_side -- python3 tools/profile_pointfusion.py 200 icp > $O/prof.txt 2>&1; grep frames/s $O/prof.txt
rocm-smi --showclocks 2>/dev/null | head -20
date
