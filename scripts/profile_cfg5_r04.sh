#!/bin/bash
# round 4: the config-5 video search (50 000 videos x 64 frames, form chosen by the probe) under rocprofv3: durations and the
# HBM-side traffic counters of its all-pairs kernel, each counter in its own pass. usage (GPU box, repo root): bash scripts/profile_cfg5_r04.sh
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_r04_cfg5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/scripts/gpu_cfg5_stages.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1; echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o $c -- $CMD > $OUT/$c.log 2>&1; echo "$c rc=$?"
done
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1; echo "sq rc=$?"
python $REPO/scripts/pmc_summary.py $OUT > $REPO/gpurun_out/r04_pmc_cfg5.txt 2>&1
grep -E "^## k_allpairs_mfma" -A 14 $REPO/gpurun_out/r04_pmc_cfg5.txt | head -60
