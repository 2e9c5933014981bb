#!/bin/bash
# rocprofv3 passes for the dominant kernels. Counters are collected in their own runs
# (--pmc with --kernel-trace only), one TCC-heavy counter group per pass.
# usage (on the GPU box, from the repo root): bash scripts/profile_pmc.sh <tag>
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $REPO/scripts/prof_driver.py > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $REPO/scripts/prof_driver.py > $OUT/stats.log 2>&1; echo "stats rc=$?"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum
find $OUT -name "*.csv" | head -30
