#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s18.txt; : > $O
run() { echo "== $1" >> $O; ( HVD_LIB_PATH=${2:-} V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 13 2>&1 | tail -2 ) >> $O; }
run "settle at 700" ""
run "settle at 900" build_tmp/libhvd_drain900.so
run "settle at 1100" build_tmp/libhvd_drain1100.so
run "settle at 700" ""
run "settle at 900" build_tmp/libhvd_drain900.so
run "settle at 1100" build_tmp/libhvd_drain1100.so
( PS=8192,2048,1024,768,512,384,256,128 timeout 900 python scripts/gpu_k2_rate_sweep.py 2>&1 | tail -8 ) >> $O
cat $O
