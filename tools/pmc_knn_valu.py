#!/usr/bin/env python3
"""Executed work of the association kernel from a rocprofv3 counter pass.

    cd /tmp && export TMPDIR=/tmp
    GS_BENCH_SHORT=1 GS_BENCH_REPEATS=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
        SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- \
        python3 bench.py --no-cpu-baseline --steps 20 --warmup 3
    python3 tools/pmc_knn_valu.py <dir>/**/*counter_collection.csv profiles/r02_pmc_knn1_loop_valu.json [grid] [first_n]

Takes the FIRST first_n (default 561 = (8 priming + 3 warm-up + 20 timed + 20 event-timed) steps x 11) launches of
knn1_loop_k with the c2 grid (300 blocks of 1024 threads = 307 200 work-items) in dispatch order -- the c2 workload
proper; bench.py's auxiliary PointFusion runs come after -- and reports per-launch means.  VALU issue utilisation = SQ_INSTS_VALU x 4 cycles (one wave64
VALU instruction holds its SIMD's issue for 4 cycles, MI355X_MICROARCH.md 'vector-instruction ISSUE cost') over
1024 SIMDs x the launch's cycles (GRBM_GUI_ACTIVE / 8 XCDs)."""
import csv, json, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 307200
first_n = int(sys.argv[4]) if len(sys.argv) > 4 else 561
per = defaultdict(lambda: defaultdict(float))
names, dur = {}, {}
HEAD = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
        "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
        "Counter_Value", "Start_Timestamp", "End_Timestamp"]
first = open(src).readline()
rows = csv.DictReader(open(src)) if first.startswith('"Correlation_Id"') else csv.DictReader(open(src), fieldnames=HEAD)
for r in rows:
    if "knn1_loop_k" not in r["Kernel_Name"] or int(r["Grid_Size"]) != grid:
        continue
    per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
keep = sorted(per)[:first_n]
per = {k: per[k] for k in keep}
names = {k: names[k] for k in keep}
n = len(per)
assert n, "no knn1_loop_k launch with grid %d in %s" % (grid, src)
mean = lambda k: sum(d.get(k, 0.0) for d in per.values()) / n
import os
out = {"commit": os.environ.get("GS_COMMIT", "unknown"),
       "source": "rocprofv3 --pmc (one pass, with --kernel-trace only) over `python3 bench.py --no-cpu-baseline --steps 20 --warmup 3`, "
                 "MI355X; tools/pmc_knn_valu.py; read from this committed profile by bench.py, NOT measured in the bench run",
       "kernel": sorted(set(x.split("(")[0] for x in names.values())), "launches": n, "grid_work_items": grid}
for k in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"):
    out[k + "_per_launch"] = round(mean(k), 1)
dur_ns = sum(dur[k] for k in keep) / n
out["launch_ns_under_the_profiler"] = round(dur_ns, 1)
cyc = dur_ns * 2.4  # cycles at the 2.4 GHz peak clock: an upper bound on the cycles available, so a lower bound on the utilisation
out["GRBM_note"] = "GRBM_GUI_ACTIVE / 8 reads high on dispatches this short (MI355X_MICROARCH.md, DVFS give-back): cycles are taken from the launch duration x 2.4 GHz instead"
if cyc > 0:
    out["launch_cycles"] = round(cyc, 1)
    out["valu_issue_utilisation"] = round(mean("SQ_INSTS_VALU") * 4.0 / (1024.0 * cyc), 4)
    out["valu_wave_instructions_per_tile"] = round(mean("SQ_INSTS_VALU") / (grid / 1024.0), 1)  # one 1024-thread block per 64-point tile
if mean("SQ_WAVE_CYCLES") > 0:
    out["wave_cycles_waiting_fraction"] = round(mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES"), 4)
    out["wave_cycles_issuing_valu_fraction"] = round(mean("SQ_ACTIVE_INST_VALU") / mean("SQ_WAVE_CYCLES"), 4)
out["reading"] = ("valu_issue_utilisation = executed VALU wave-instructions x 4 issue cycles / (1024 SIMDs x launch cycles): the share of the "
                  "chip's vector issue slots the launch fills.  The kernel is bound by dependent-launch latency, the O(1) step on one wave and "
                  "memory latency in the search's box tests, not by VALU throughput or HBM bandwidth.")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
