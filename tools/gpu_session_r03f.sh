#!/bin/bash
# GPU session r03f: window centres from the previous launch's cells (one dependent load less), camera constants in LDS.
set -o pipefail
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
echo "== grid tests"; date
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/tests_a.log 2>&1; echo "rc=$?"
tail -8 $O/tests_a.log
for n in 150 6; do
  GS_GRID_MODE=2 timeout -k 10 200 python tools/knn_diag_long.py $n > $O/diag_m2_$n.txt 2>&1; echo "diag $n rc=$?"; sed -n 3,12p $O/diag_m2_$n.txt
done
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
timeout -k 10 200 python tools/batch_scaling.py 40 4 icp 2>&1 | tail -1
echo "== pf200 icp under rocprofv3 (default policy)"; date
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; echo "rc=$?"; grep frames/s $O/pf200_prof.txt
date
