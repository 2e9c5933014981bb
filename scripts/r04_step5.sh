#!/bin/bash
set -u
mkdir -p gpurun_out
( for lib in "" build_tmp/libhvd_qabl2.so build_tmp/libhvd_qabl3.so; do echo "lib=$lib"; HVD_LIB_PATH=$lib V=16000 timeout 600 python scripts/gpu_k2_structured.py 15; done 2>&1 ) > gpurun_out/r04_s5_abl.txt
( timeout 600 python scripts/gpu_k2_uniform.py 9 15 2>&1 ) > gpurun_out/r04_s5_uniform.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "k2 or k3" 2>&1 | tail -8 ) > gpurun_out/r04_s5_pytest.txt
( timeout 900 python scripts/gpu_fuzz_k2.py 16 3000 2>&1 | tail -5 ) > gpurun_out/r04_s5_fuzz.txt
tail -n 30 gpurun_out/r04_s5_*.txt
