#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s21.txt; : > $O
( PS=16384,4096,1024,512,256,192,160,128,64 timeout 900 python scripts/gpu_k2_rate_sweep.py 2>&1 | tail -9 ) >> $O
for v in 17 18 19; do ( python scripts/gpu_k2_missing.py $v 2>&1 | tail -1 ) >> $O; done
( timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_group.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) >> $O
( timeout 900 python scripts/gpu_fuzz_k2.py 40 7000 2>&1 | tail -3 ) >> $O
cat $O
