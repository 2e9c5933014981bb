#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s30.txt; : > $O
run() { echo "== $1" >> $O; ( HVD_LIB_PATH=${2:-} V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 2>&1 | tail -1 ) >> $O; }
( python scripts/gpu_k2_missing.py 18 2>&1 | tail -1 ) >> $O
for i in 1 2 3; do
run "eight rows at once" ""
run "chunk by chunk" build_tmp/libhvd_nopair.so
done
( timeout 900 python scripts/gpu_fuzz_k2.py 20 14000 2>&1 | tail -2 ) >> $O
cat $O
