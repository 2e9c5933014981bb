#!/bin/bash
# round 4: PMC breakdown of the pair-queue form (15) against the register form (12) on structured frame hashes, and of
# the fetch form (9) on uniform hashes. usage (GPU box, repo root): bash scripts/profile_k2_r04.sh [tags...]
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_r04_k2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
G2="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD"
for tag in ${@:-structured18 structured15 structured12 uniform9}; do
  case $tag in
    structured*) CMD="python $REPO/scripts/gpu_k2_structured.py ${tag#structured}"; export V=16000;;
    uniform*) CMD="python $REPO/scripts/prof_driver.py 920000 ${tag#uniform}";;
  esac
  mkdir -p $OUT/$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/stats -o stats -- $CMD > $OUT/$tag/stats.log 2>&1
  rocprofv3 --kernel-trace --pmc $G1 --output-format csv -d $OUT/$tag/g1 -o g1 -- $CMD > $OUT/$tag/g1.log 2>&1
  rocprofv3 --kernel-trace --pmc $G2 --output-format csv -d $OUT/$tag/g2 -o g2 -- $CMD > $OUT/$tag/g2.log 2>&1
  rocprofv3 --kernel-trace --pmc $G3 --output-format csv -d $OUT/$tag/g3 -o g3 -- $CMD > $OUT/$tag/g3.log 2>&1
  python $REPO/scripts/pmc_summary.py $OUT/$tag 2>&1 | awk '/^## k_allpairs_mfma/{p=1} /^## /{if(!/k_allpairs_mfma/)p=0} p' > $REPO/gpurun_out/r04_pmc_k2_$tag.txt
  tail -3 $OUT/$tag/g3.log >> $REPO/gpurun_out/r04_pmc_k2_$tag.log
done
echo done
