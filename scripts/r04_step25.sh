#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s25.txt; : > $O
( timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_group.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) >> $O
( timeout 600 python scripts/gpu_k2_uniform.py 9 13 9 13 2>&1 | tail -4 ) >> $O
( V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 13 2>&1 | tail -2 ) >> $O
( timeout 900 python scripts/gpu_fuzz_k2.py 20 12000 2>&1 | tail -2 ) >> $O
cat $O
