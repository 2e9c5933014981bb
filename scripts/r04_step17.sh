#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s17.txt; : > $O
run() { echo "== $1" >> $O; ( HVD_LIB_PATH=${2:-} V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 17 9 2>&1 | tail -3 ) >> $O; }
run default ""
run "QABL=8 (nothing pushed)" build_tmp/libhvd_qabl8.so
run "QABL=10 (nothing pushed, fill levels not read)" build_tmp/libhvd_qabl10.so
run default ""
cat $O
