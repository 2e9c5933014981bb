#!/bin/bash
# Round-2 profile set (run on the GPU box from the repo root): bash scripts/profile_r02.sh
#   1. rocprofv3 --kernel-trace --stats of bench.py itself (same command as the bench line, short legs)
#   2. PMC passes of the dominant kernels (scripts/profile_pmc.sh: counters in their own runs)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT/r02_stats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02_stats -o bench -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustain-seconds 2 > $OUT/r02_bench_prof.json 2> $OUT/r02_bench_prof.err
echo "bench under rocprof rc=$?"
find $OUT/r02_stats -name "*kernel_stats.csv" | head -3
cd $REPO && bash scripts/profile_pmc.sh r02
python scripts/pmc_summary.py gpurun_out/pmc_r02 > gpurun_out/r02_pmc_summary.txt 2>&1
tail -5 gpurun_out/r02_pmc_summary.txt
