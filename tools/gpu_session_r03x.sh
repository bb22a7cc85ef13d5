#!/bin/bash
# GPU session r03x: which piece of corr_pass1_k / merge_corr_k the time belongs to (tools/corr_variants.py)
set -o pipefail
O=gpurun_out/r03x; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python tools/corr_variants.py 100 > $O/variants_events.txt 2>&1; echo "rc=$?"; cat $O/variants_events.txt | tail -80
date
