#!/usr/bin/env python3
"""Memory-side traffic of the association kernel from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; each with
--kernel-trace only) over `python3 bench.py --no-cpu-baseline --steps 20 --warmup 3`.

    python3 tools/pmc_knn_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [grid] [first_n]

Takes the first first_n launches of knn1_loop_k with the c2 grid in dispatch order (as tools/pmc_knn_valu.py does)."""
import csv, json, os, sys
from collections import defaultdict

HEAD = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
        "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
        "Counter_Value", "Start_Timestamp", "End_Timestamp"]


def per_launch(src, counter, grid, first_n):
    first = open(src).readline()
    rows = csv.DictReader(open(src)) if first.startswith('"Correlation_Id"') else csv.DictReader(open(src), fieldnames=HEAD)
    per = defaultdict(float)
    for r in rows:
        if "knn1_loop_k" in r["Kernel_Name"] and int(r["Grid_Size"]) == grid and r["Counter_Name"] == counter:
            per[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    keep = sorted(per)[:first_n]
    assert keep, "no knn1_loop_k launch with grid %d and counter %s in %s" % (grid, counter, src)
    v = [per[k] for k in keep]
    return len(v), sum(v) / len(v), min(v), max(v)


fsrc, wsrc, dst = sys.argv[1], sys.argv[2], sys.argv[3]
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 307200
first_n = int(sys.argv[5]) if len(sys.argv) > 5 else 561
nf, f_mean, f_min, f_max = per_launch(fsrc, "FETCH_SIZE", grid, first_n)
nw, w_mean, _, _ = per_launch(wsrc, "WRITE_SIZE", grid, first_n)
ns = 19200  # source points of the c2 step's capacity (160 x 120)
out = {"commit": os.environ.get("GS_COMMIT", "unknown"),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) over `python3 bench.py "
                 "--no-cpu-baseline --steps 20 --warmup 3`, MI355X; tools/pmc_knn_traffic.py",
       "kernel": "knn1_loop_k, c2 grid (%d work-items), first %d launches in dispatch order" % (grid, nf),
       "FETCH_SIZE_KiB_mean": round(f_mean, 1), "FETCH_SIZE_KiB_min": round(f_min, 1), "FETCH_SIZE_KiB_max": round(f_max, 1),
       "WRITE_SIZE_KiB_mean": round(w_mean, 1),
       "bytes_fetched_per_launch_raw": round(f_mean * 1024), "bytes_fetched_per_launch_x2_corrected": round(2 * f_mean * 1024),
       "bytes_written_per_launch": round(w_mean * 1024),
       "algorithmic_bytes_per_launch": 40 * ns,
       "reading": "units: KiB per launch (FETCH_SIZE / WRITE_SIZE count 1 KiB units on gfx950; the x2 correction is the one calibrated on a "
                  "dword-x3 stream in r01_pmc_traffic.json, not re-calibrated for this access pattern).  Compare with "
                  "profiles/r01t_pmc_knn1_loop.json (round 1: 1 842 KiB fetched, 395 KiB written per launch)."}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
