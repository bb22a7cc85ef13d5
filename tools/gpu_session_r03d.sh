#!/bin/bash
# GPU session r03d: radius-2 windows on sparse targets; grid search at every density vs the default policy; tile sizes.
set -o pipefail
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
echo "== grid tests"; date
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -s -m gpu -k "grid_search or tile_points or straggler or dense_regime or c3_64 or fixture_full" > $O/tests_a.log 2>&1; echo "rc=$?"
grep -E "passed|failed|pose rel err|map size differs|overflow|Error" $O/tests_a.log | tail -24
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
run grid_always_r1 GS_GRID_MODE=2 GS_GRID_RADIUS=1
run grid_always_tile64 GS_GRID_MODE=2 GS_TILE_POINTS=64
run grid_always_tile38 GS_GRID_MODE=2 GS_TILE_POINTS=38
run grid_always_r2below6 GS_GRID_MODE=2 GS_GRID_R2_BELOW=6
date
