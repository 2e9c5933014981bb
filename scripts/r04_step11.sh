#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s11.txt; : > $O
for i in 1 2; do
echo "== no stagger" >> $O; ( HVD_LIB_PATH=build_tmp/libhvd_nostagger.so V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 15 19 2>&1 | tail -3 ) >> $O
echo "== stagger" >> $O; ( V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 15 19 2>&1 | tail -3 ) >> $O
done
( python scripts/gpu_k2_missing.py 18 2>&1 | tail -2 ) >> $O
cat $O
