#!/bin/bash
# Panel-mark queue form (variant 17) against the group-mask queue (15), register (12) and fetch (9) forms, same box.
set -u
mkdir -p gpurun_out
( python scripts/gpu_k2_missing.py 17 2>&1 | tail -5 ) > gpurun_out/r04_s8_missing.txt
for i in 1 2; do
( V=16000 timeout 600 python scripts/gpu_k2_structured.py 17 15 12 2>&1 | tail -3 ) > gpurun_out/r04_s8_structured_$i.txt
done
( timeout 600 python scripts/gpu_k2_uniform.py 9 17 15 2>&1 | tail -4 ) > gpurun_out/r04_s8_uniform.txt
( timeout 900 python scripts/gpu_fuzz_k2.py 40 4000 2>&1 | tail -5 ) > gpurun_out/r04_s8_fuzz.txt
tail -n 30 gpurun_out/r04_s8_*.txt
