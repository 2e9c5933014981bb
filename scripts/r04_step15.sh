#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s15.txt; : > $O
run() { echo "== $1" >> $O; ( HVD_LIB_PATH=${2:-} V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 17 19 15 2>&1 | tail -4 ) >> $O; }
for v in 17 18 19; do ( python scripts/gpu_k2_missing.py $v 2>&1 | tail -1 ) >> $O; done
for i in 1 2; do
run "full check in the filter" ""
run "without" build_tmp/libhvd_nofull.so
done
( timeout 600 python scripts/gpu_k2_uniform.py 9 18 17 2>&1 | tail -3 ) >> $O
cat $O
