#!/bin/bash
set -u
mkdir -p gpurun_out
( python scripts/gpu_k2_missing.py 15 2>&1 | tail -5 ) > gpurun_out/r04_s6_missing.txt
( V=16000 timeout 600 python scripts/gpu_k2_structured.py 15 16 12 13 2>&1 ) > gpurun_out/r04_s6_structured.txt
( timeout 600 python scripts/gpu_k2_uniform.py 9 15 16 13 2>&1 ) > gpurun_out/r04_s6_uniform.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "k2 or k3" 2>&1 | tail -8 ) > gpurun_out/r04_s6_pytest.txt
( timeout 900 python scripts/gpu_fuzz_k2.py 40 3000 2>&1 | tail -5 ) > gpurun_out/r04_s6_fuzz.txt
tail -n 30 gpurun_out/r04_s6_*.txt
