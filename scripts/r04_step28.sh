#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s28.txt; : > $O
for c in 8192 16384 32768 65536; do
echo "== col chunk max $c" >> $O
( CHUNK=$c V=16000 timeout 600 python scripts/gpu_k2_structured.py 13 2>&1 | tail -1 ) >> $O
( CHUNK=$c timeout 600 python scripts/gpu_k2_uniform.py 13 2>&1 | tail -1 ) >> $O
( CHUNK=$c timeout 600 python scripts/gpu_cfg5_stages.py 2>&1 | grep match_videos | tail -1 ) >> $O
done
cat $O
