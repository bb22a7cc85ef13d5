#!/bin/bash
# GPU session r03o: counter passes of the shipped tree (the bench line must not mix live timings with counters of
# an older kernel): VALU view of the c2 association kernel, HBM traffic of J at 2^24 points (separate --pmc passes,
# --kernel-trace only), then the rocprofv3 summaries of bench.py and of the 200-frame runs.
set -o pipefail
R=$PWD
O=$R/gpurun_out/r03o; mkdir -p $O
export GS_COMMIT=336d37d
cd /tmp && export TMPDIR=/tmp
echo "== pmc valu"; date
GS_BENCH_SHORT=1 GS_BENCH_REPEATS=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/pmc_valu.log 2>&1; echo "rc=$?"
python3 $R/tools/pmc_knn_valu.py $(ls $O/pmc_valu/*/*counter_collection.csv | head -1) $O/r03_pmc_knn1_loop_valu.json 307200 561 | tail -12
echo "== pmc fetch / write"; date
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/tools/pmc_traffic.py > $O/pmc_f.log 2>&1; echo "rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/tools/pmc_traffic.py > $O/pmc_w.log 2>&1; echo "rc=$?"
python3 $R/tools/pmc_traffic_summary.py $O/pmc_f $O/pmc_w $O/r03_pmc_traffic.json | tail -20
echo "== rocprofv3 summaries"; date
cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err; echo "rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pf200 -- python3 tools/profile_pointfusion.py 200 icp > $O/pf200_prof.txt 2>&1; grep frames/s $O/pf200_prof.txt
find $O -name "*kernel_trace.csv" -size +30M -delete
find $O -name "*counter_collection.csv" -size +30M -delete
date
