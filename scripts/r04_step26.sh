#!/bin/bash
set -u
mkdir -p gpurun_out
O=$(pwd)/gpurun_out/r04_s26.txt; : > $O
R=$(pwd)
( timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) >> $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr1 /tmp/tr2
V=16000 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o t -- python $R/scripts/gpu_k2_structured.py 13 > /dev/null 2>&1
python $R/scripts/trace_gaps.py /tmp/tr1 >> $O
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr2 -o t -- python $R/scripts/gpu_k2_uniform.py 13 > /dev/null 2>&1
python $R/scripts/trace_gaps.py /tmp/tr2 >> $O
cd $R
( timeout 600 python scripts/gpu_k2_uniform.py 9 13 9 13 2>&1 | tail -4 ) >> $O
( V=16000 timeout 600 python scripts/gpu_k2_structured.py 18 13 18 13 2>&1 | tail -4 ) >> $O
cat $O
