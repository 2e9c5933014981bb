#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s19.txt; : > $O
run() { echo "== $1" >> $O; ( HVD_LIB_PATH=${2:-} V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 2>&1 | tail -1 ) >> $O; }
for i in 1 2; do
run "E=3, settle at 700" ""
run "E=2, settle at 500" build_tmp/libhvd_e2.so
run "E=4, settle at 1000" build_tmp/libhvd_e4.so
run "E=5, settle at 1250 (capacity-bound)" build_tmp/libhvd_e5.so
done
cat $O
