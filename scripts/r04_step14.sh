#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s14.txt; : > $O
run() { echo "== $1" >> $O; ( HVD_LIB_PATH=${2:-} V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 17 15 2>&1 | tail -3 ) >> $O; }
for i in 1 2; do
run "rotation + split" ""
run "no rotation" build_tmp/libhvd_norot.so
run "no split" build_tmp/libhvd_nosplit.so
run "neither" build_tmp/libhvd_neither.so
done
( timeout 600 python scripts/gpu_k2_uniform.py 9 18 2>&1 | tail -2 ) >> $O
( HVD_LIB_PATH=build_tmp/libhvd_nosplit.so timeout 600 python scripts/gpu_k2_uniform.py 9 18 2>&1 | tail -2 ) >> $O
cat $O
