#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s13.txt; : > $O
for v in 17 18 19; do ( python scripts/gpu_k2_missing.py $v 2>&1 | tail -1 ) >> $O; done
for i in 1 2; do ( V=16000 timeout 600 python scripts/gpu_k2_structured.py 17 18 19 15 2>&1 | tail -4 ) >> $O; done
( timeout 600 python scripts/gpu_k2_uniform.py 9 17 18 19 2>&1 | tail -4 ) >> $O
( timeout 900 python scripts/gpu_fuzz_k2.py 30 6000 2>&1 | tail -3 ) >> $O
cat $O
