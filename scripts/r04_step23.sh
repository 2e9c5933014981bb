#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s23.txt; : > $O
( timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) >> $O
( timeout 600 python scripts/gpu_k2_uniform.py 9 13 9 13 2>&1 | tail -4 ) >> $O
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python $OLDPWD/scripts/gpu_k2_uniform.py 13 > /dev/null 2>&1; grep -E "k_prefilter_probe|k_probe_decide" /tmp/pp/*/pp_kernel_stats.csv /tmp/pp/pp_kernel_stats.csv 2>/dev/null | cut -c1-200 ) >> $O
cat $O
