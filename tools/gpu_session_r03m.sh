#!/bin/bash
# GPU session r03m: the host kept within two frames of the map counts it knows (arena bounds): clean-run forward / backward.
set -o pipefail
O=gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 python tools/profile_pointfusion.py 200 icp 2>&1 | tail -2
timeout -k 10 200 python tools/profile_pointfusion.py 200 gradicp 2>&1 | tail -1
timeout -k 10 300 python tools/fwd_bwd_c3.py 200 gradicp 2>&1 | tail -3
echo "== all gpu tests"; date
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -5 $O/gpu_tests.log
echo "== full bench"; date
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -3 $O/bench.err
python - <<P
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("c2", j["value"], j["ms_per_step"], "assoc us", round(1e3*j["roofline_timed_region"]["avg_launch_ms"],2))
print("roofline", j["roofline"]["frac"], "real assoc", j.get("roofline_real_associations"))
a=j["aux"]; print("aux30", a["pointfusion_c3_forward_fps"], a["pointfusion_c3_forward_fps_stepwise_api"], a["pointfusion_c3_gradicp_fwd_bwd_fps"])
print("fusion", {k:v for k,v in a["fusion_update_hbm_view"].items() if k!="note"})
print("c3", {k:v for k,v in a["pointfusion_c3_200_frames"].items() if k!="note"})
P
date
