#!/bin/bash
set -u
mkdir -p gpurun_out
( V=16000 timeout 600 python scripts/gpu_k2_structured.py 12 15 12 15 2>&1 ) > gpurun_out/r04_s2_structured.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "k1 or k2 or pdq or hash" 2>&1 | tail -8 ) > gpurun_out/r04_s2_pytest.txt
( timeout 300 python scripts/gpu_k1_time2.py 2>&1 ) > gpurun_out/r04_s2_k1.txt
tail -n 30 gpurun_out/r04_s2_*.txt
