#!/bin/bash
# PMC stall breakdown of the register form (variant 12) on structured frame hashes and of the fetch form (9) / register
# form (12) on uniform hashes. usage (GPU box, repo root): bash scripts/profile_k2_structured_stalls.sh
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_r03_k2_stalls; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for tag in structured12 uniform12 uniform9; do
  case $tag in
    structured12) CMD="python $REPO/scripts/gpu_k2_structured.py 12"; export V=16000;;
    uniform12) CMD="python $REPO/scripts/prof_driver.py 920000 12";;
    uniform9) CMD="python $REPO/scripts/prof_driver.py 920000 9";;
  esac
  mkdir -p $OUT/$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/stats -o stats -- $CMD > $OUT/$tag/stats.log 2>&1
  rocprofv3 --kernel-trace --pmc $G1 --output-format csv -d $OUT/$tag/g1 -o g1 -- $CMD > $OUT/$tag/g1.log 2>&1
  rocprofv3 --kernel-trace --pmc $G2 --output-format csv -d $OUT/$tag/g2 -o g2 -- $CMD > $OUT/$tag/g2.log 2>&1
  python $REPO/scripts/pmc_summary.py $OUT/$tag > $REPO/gpurun_out/r03_pmc_k2_stalls_$tag.txt 2>&1
done
echo done
