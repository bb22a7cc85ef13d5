#!/bin/bash
# GPU session r05i: memory-side traffic of the rebuilt association kernel (FETCH_SIZE / WRITE_SIZE passes over the short bench)
set -o pipefail
R=$PWD
O=$R/gpurun_out/r05i; mkdir -p $O
export GS_COMMIT=424e811
export TMPDIR=/tmp
cd /tmp
GS_BENCH_SHORT=1 GS_BENCH_REPEATS=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/pmc_f.log 2>&1; echo "rc=$?"
GS_BENCH_SHORT=1 GS_BENCH_REPEATS=1 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/pmc_w.log 2>&1; echo "rc=$?"
python3 $R/tools/pmc_knn_traffic.py $(ls $O/pmc_f/*/*counter_collection.csv | head -1) $(ls $O/pmc_w/*/*counter_collection.csv | head -1) $O/r05_pmc_knn1_loop_traffic.json | grep -E "commit|KiB|bytes"
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*counter_collection.csv" -size +20M -delete
date
