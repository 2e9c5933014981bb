#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s20.txt; : > $O
run() { echo "== $1" >> $O; ( HVD_LIB_PATH=${2:-} V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 2>&1 | tail -1 ) >> $O; ( HVD_LIB_PATH=${2:-} PS=1024,512,384,256,192,128 timeout 900 python scripts/gpu_k2_rate_sweep.py 2>&1 | tail -6 | cut -c1-150 ) >> $O; }
run "16 lanes" ""
run "32 lanes" build_tmp/libhvd_pl32.so
run "48 lanes" build_tmp/libhvd_pl48.so
cat $O
