#!/bin/bash
# GPU session r05a: everything the judged files come from, on ONE tree: smoke, the GPU suite, counter passes (VALU view of
# the c2 association kernel, HBM traffic of J at 2^24), rocprofv3 summaries of bench.py and of the 200-frame runs, a two-rank
# rehearsal of bench.py on the one card (gloo: RCCL refuses two ranks on one device; ranks_seen), the full bench line with the CPU baseline.
set -o pipefail
R=$PWD
O=$R/gpurun_out/r05a; mkdir -p $O
export GS_COMMIT=ee311fe
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== all gpu tests"; date
timeout -k 10 800 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
cd /tmp
echo "== pmc valu"; date
GS_BENCH_SHORT=1 GS_BENCH_REPEATS=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/pmc_valu.log 2>&1; echo "rc=$?"
python3 $R/tools/pmc_knn_valu.py $(ls $O/pmc_valu/*/*counter_collection.csv | head -1) $O/r03_pmc_knn1_loop_valu.json 307200 561 | grep -E "commit|valu_issue|waiting|launch_ns"
echo "== pmc fetch / write"; date
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/tools/pmc_traffic.py > $O/pmc_f.log 2>&1; echo "rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/tools/pmc_traffic.py > $O/pmc_w.log 2>&1; echo "rc=$?"
python3 $R/tools/pmc_traffic_summary.py $O/pmc_f $O/pmc_w $O/r03_pmc_traffic.json | grep -E "commit|hbm_bytes|fetch_correction"
echo "== rocprofv3 summaries"; date
cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err; echo "rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; grep frames/s $O/pf200_prof.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fb -- python3 tools/fwd_bwd_c3.py 200 gradicp > $O/fb_prof.txt 2>&1; tail -1 $O/fb_prof.txt
echo "== two ranks on one card"; date
GS_DIST_BACKEND=gloo GS_BENCH_SHORT=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 > $O/bench_2rank.json 2> $O/bench_2rank.err; echo "rc=$?"; cut -c1-260 $O/bench_2rank.json
echo "== full bench"; date
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?"; tail -2 $O/bench.err
python - <<P
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("c2", j["value"], j["ms_per_step"], j["repeats_ms_per_step"], "assoc us", round(1e3*j["roofline_timed_region"]["avg_launch_ms"],2))
print("roofline", j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"]["traffic_source"][:60], "| real", j["roofline_real_associations"]["frac"])
print("valu view", j["roofline_timed_region"]["valu_view"].get("commit"), j["roofline_timed_region"]["valu_view"].get("valu_issue_utilisation"))
print("cpu", {k:v for k,v in j["cpu_baseline"].items() if k in ("value","cores","kind")}, j["cpu_baseline"]["nn_single_thread"]["seconds_per_search"], j["cpu_baseline"]["c3_pointfusion"]["seconds_per_frame"])
a=j["aux"]; print("aux30", a["pointfusion_c3_forward_fps"], a["pointfusion_c3_forward_fps_stepwise_api"], a["pointfusion_c3_gradicp_fwd_bwd_fps"])
print("fusion", {k:v for k,v in a["fusion_update_hbm_view"].items() if k!="note"})
print("c3", {k:v for k,v in a["pointfusion_c3_200_frames"].items() if k!="note"})
P
find $O -name "*kernel_trace.csv" -size +30M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
date
