set -e
O=gpurun_out/r02_bb; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "grid_search or tile_points or knn" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
for pm in 8 0 16 8; do echo "== pair_max $pm"; GS_PAIR_MAX=$pm python tools/profile_pointfusion.py 200 icp; done > $O/pf.log 2>&1
grep -v amdgpu.ids $O/pf.log
echo "== gradicp pm 8"; python tools/profile_pointfusion.py 200 gradicp 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bb -- python3 $GRAFT_REPO_ROOT/tools/profile_pointfusion.py 200 icp > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT && cp $(find /tmp/prof_bb -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv && python tools/trace_assoc.py $(find /tmp/prof_bb -name "*kernel_trace.csv" | head -1) 11 > $O/assoc_by_position.txt 2>&1; head -4 $O/kernel_stats.csv | cut -c1-60,190-260; cat $O/assoc_by_position.txt
