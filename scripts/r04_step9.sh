#!/bin/bash
# Ablations of the panel-mark queue form (variant 17) on structured hashes, same box (timing-only libs: wrong results).
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s9.txt; : > $O
run() { echo "== $1" >> $O; ( HVD_LIB_PATH=${2:-} V=16000 timeout 600 python scripts/gpu_k2_structured.py 17 2>&1 | tail -1 ) >> $O; }
run default ""
run "QABL=8 (nothing pushed)" build_tmp/libhvd_qabl8.so
run "QABL=3 (pushed, dropped)" build_tmp/libhvd_qabl3.so
run "QABL=6 (drain = call only)" build_tmp/libhvd_qabl6.so
run default ""
( python scripts/gpu_k2_missing.py 17 2>&1 | tail -2 ) >> $O
( timeout 600 python scripts/gpu_k2_uniform.py 9 17 2>&1 | tail -2 ) >> $O
cat $O
