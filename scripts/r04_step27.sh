#!/bin/bash
set -u
mkdir -p gpurun_out
O=$(pwd)/gpurun_out/r04_s27.txt; : > $O
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr3
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr3 -o t -- python $R/scripts/gpu_cfg5_stages.py > /tmp/cfg5.log 2>&1
tail -8 /tmp/cfg5.log >> $O
python $R/scripts/trace_cfg5.py /tmp/tr3 >> $O
cat $O
