#!/bin/bash
# GPU session r03i: staging requests its first loads before the folded step's partial rows; band loads independent.
set -o pipefail
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
echo "== grid tests"; date
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "grid_search or tile_points or straggler or dense_regime or reproducible or streamed_arena" > $O/tests_a.log 2>&1; echo "rc=$?"
tail -3 $O/tests_a.log
GS_GRID_MODE=2 timeout -k 10 200 python tools/knn_diag_long.py 150 > $O/diag_m2_150.txt 2>&1; sed -n 3,12p $O/diag_m2_150.txt
GS_GRID_MODE=2 timeout -k 10 200 python tools/knn_diag_long.py 6 > $O/diag_m2_6.txt 2>&1; sed -n 3,12p $O/diag_m2_6.txt
run() {  # label, env...
  local label=$1; shift
  env "$@" GS_BENCH_SHORT=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_$label.json 2> $O/bench_$label.err
  python - <<P
import json
j=json.loads(open("$O/bench_$label.json").read().strip().splitlines()[-1])
print("$label", "c2 ms/step", j["ms_per_step"], "fps", j["value"], "assoc us", round(1e3*j["roofline_timed_region"]["avg_launch_ms"],2), "pf30", j["aux"]["pointfusion_c3_forward_fps"], "fwd+bwd30", j["aux"]["pointfusion_c3_gradicp_fwd_bwd_fps"])
P
  env "$@" timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -1
  env "$@" timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp 2>&1 | tail -1
}
run default GS_X=0
run grid_always GS_GRID_MODE=2
echo "== pf200 icp under rocprofv3, default and grid always"; date
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; grep frames/s $O/pf200_prof.txt
GS_GRID_MODE=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200_m2 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof_m2.txt 2>&1; grep frames/s $O/pf200_prof_m2.txt
date
