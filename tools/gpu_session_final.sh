#!/bin/bash
# Final checks of a tree: smoke, the whole GPU suite, the full bench line (with the CPU baseline).
set -o pipefail
O=gpurun_out/${1:-r03_final}; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== all gpu tests"; date
timeout -k 10 800 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -4 $O/gpu_tests.log
echo "== full bench"; date
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -2 $O/bench.err
python - <<P
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("c2", j["value"], j["ms_per_step"], j["repeats_ms_per_step"], "assoc us", round(1e3*j["roofline_timed_region"]["avg_launch_ms"],2))
print("roofline", j["roofline"]["frac"], j["roofline"]["traffic"], "| real", j["roofline_real_associations"]["frac"])
print("cpu", {k:v for k,v in j["cpu_baseline"].items() if k in ("value","cores","kind")}, j["cpu_baseline"]["nn_single_thread"]["seconds_per_search"], j["cpu_baseline"]["c3_pointfusion"]["seconds_per_frame"])
a=j["aux"]; print("aux30", a["pointfusion_c3_forward_fps"], a["pointfusion_c3_forward_fps_stepwise_api"], a["pointfusion_c3_gradicp_fwd_bwd_fps"])
print("fusion", {k:v for k,v in a["fusion_update_hbm_view"].items() if k!="note"})
print("c3", {k:v for k,v in a["pointfusion_c3_200_frames"].items() if k!="note"})
print("assoc sizes", {k:(v["ms_per_association"] if isinstance(v,dict) else None) for k,v in a["association_other_sizes"].items()})
P
date
