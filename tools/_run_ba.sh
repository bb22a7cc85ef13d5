set -e
O=gpurun_out/r02_ba; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "grid_search or tile_points or knn" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -3 $O/t.log
for tp in 64 0 44; do echo "== tp $tp"; GS_TILE_POINTS=$tp python tools/profile_pointfusion.py 200 icp; done > $O/pf.log 2>&1
cat $O/pf.log | grep -v amdgpu.ids
for tp in 64 0; do GS_TILE_POINTS=$tp python bench.py --no-cpu-baseline --steps 50 --warmup 5 > $O/bench_$tp.log 2>$O/bench_$tp.err; python - <<PY
import json
d=json.loads(open("$O/bench_$tp.log").read().strip().splitlines()[-1])
print("tp $tp value", d["value"], d["repeats_ms_per_step"], "assoc us", d["roofline_timed_region"]["avg_launch_ms"], "c3_200", d["aux"].get("pointfusion_c3_200_frames"))
PY
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ba -- python3 $GRAFT_REPO_ROOT/tools/profile_pointfusion.py 200 icp > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT && cp $(find /tmp/prof_ba -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv && python tools/trace_assoc.py $(find /tmp/prof_ba -name "*kernel_trace.csv" | head -1) 11 > $O/assoc_by_position.txt 2>&1; head -4 $O/kernel_stats.csv | cut -c1-200; cat $O/assoc_by_position.txt
