#!/bin/bash
# Same-box A/B: HEAD library (build_tmp/libhvd_head.so) against the working tree, structured (config-5) and uniform hashes.
set -u
mkdir -p gpurun_out
for i in 1 2; do
( HVD_LIB_PATH=build_tmp/libhvd_head.so V=16000 timeout 600 python scripts/gpu_k2_structured.py 15 12 2>&1 | tail -2 ) > gpurun_out/r04_s7_head_$i.txt
( V=16000 timeout 600 python scripts/gpu_k2_structured.py 15 12 2>&1 | tail -2 ) > gpurun_out/r04_s7_new_$i.txt
done
( python scripts/gpu_k2_missing.py 15 2>&1 | tail -5 ) > gpurun_out/r04_s7_missing.txt
( timeout 900 python scripts/gpu_fuzz_k2.py 30 3000 2>&1 | tail -5 ) > gpurun_out/r04_s7_fuzz.txt
tail -n 30 gpurun_out/r04_s7_*.txt
