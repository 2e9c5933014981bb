#!/bin/bash
set -u
mkdir -p gpurun_out
( for lib in "" build_tmp/libhvd_qabl3.so build_tmp/libhvd_qabl4.so; do echo "lib=$lib"; HVD_LIB_PATH=$lib V=16000 timeout 600 python scripts/gpu_k2_structured.py 15; done 2>&1 ) > gpurun_out/r04_s4_abl.txt
cat gpurun_out/r04_s4_abl.txt
