#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s16.txt; : > $O
run() { echo "== $1" >> $O; ( HVD_LIB_PATH=${2:-} V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 2>&1 | tail -1 ) >> $O; }
run default ""
run "QABL=8 (nothing pushed)" build_tmp/libhvd_qabl8.so
run "QABL=6 (settlement = call only)" build_tmp/libhvd_qabl6.so
run "QABL=9 (fetches only)" build_tmp/libhvd_qabl9.so
run default ""
run "settle at 500" build_tmp/libhvd_drain500.so
run "settle at 1000" build_tmp/libhvd_drain1000.so
run default ""
( timeout 600 python scripts/gpu_k2_uniform.py 9 2>&1 | tail -1 ) >> $O
cat $O
