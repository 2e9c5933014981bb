#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s29.txt; : > $O
( python scripts/gpu_k2_missing.py 20 2>&1 | tail -1 ) >> $O
( python scripts/gpu_k2_missing.py 18 2>&1 | tail -1 ) >> $O
for i in 1 2 3; do ( V=16000 timeout 600 python scripts/gpu_k2_structured.py 20 18 2>&1 | tail -2 ) >> $O; done
( timeout 600 python scripts/gpu_k2_uniform.py 9 20 18 2>&1 | tail -3 ) >> $O
( timeout 900 python scripts/gpu_fuzz_k2.py 40 13000 2>&1 | tail -2 ) >> $O
cat $O
