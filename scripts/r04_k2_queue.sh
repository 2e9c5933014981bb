#!/bin/bash
# round 4: first run of the pair-queue form (variant 15): parity, structured and uniform A/B, fuzz
set -u
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "k2 or k3 or config5 or match" 2>&1 | tail -15 ) > gpurun_out/r04_q_pytest.txt
( V=16000 timeout 600 python scripts/gpu_k2_structured.py 12 15 13 9 2>&1 ) > gpurun_out/r04_q_structured.txt
( timeout 600 python scripts/gpu_k2_uniform.py 9 15 13 12 2>&1 ) > gpurun_out/r04_q_uniform.txt
( timeout 900 python scripts/gpu_fuzz_k2.py 80 1000 2>&1 | tail -20 ) > gpurun_out/r04_q_fuzz.txt
tail -n 30 gpurun_out/r04_q_*.txt
