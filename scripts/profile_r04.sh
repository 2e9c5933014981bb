#!/bin/bash
# Round-4 profile set (run on the GPU box from the repo root): bash scripts/profile_r04.sh
#   1. rocprofv3 --kernel-trace --stats of bench.py itself (same command as the bench line, short legs)
#   2. PMC passes of the dominant kernels (scripts/profile_pmc.sh: counters in their own runs)
#   3. 64x64 hash kernel at 10k / 400k frames, 512x512 front-end (k_down512w) incl. the LDS counters
#   4. structured-data K2 (config-5 generator): pair-queue form vs register form, stall and instruction counters (profile_k2_r04.sh)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT/r04_stats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_stats -o bench -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $OUT/r04_bench_prof.json 2> $OUT/r04_bench_prof.err
echo "bench under rocprof rc=$?"
find $OUT/r04_stats -name "*kernel_stats.csv" | head -3
cd $REPO && bash scripts/profile_pmc.sh r04
python scripts/pmc_summary.py gpurun_out/pmc_r04 > gpurun_out/r04_pmc_summary.txt 2>&1
tail -5 gpurun_out/r04_pmc_summary.txt
cd $REPO
for N in 10000 400000; do
  O2=$OUT/pmc_r04_k1_$N; mkdir -p $O2
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $O2/stats -o stats -- python $REPO/scripts/prof_k1.py $N > $O2/stats.log 2>&1
    for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
      set -- $grp; name=$1; shift
      rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O2/$name -o $name -- python $REPO/scripts/prof_k1.py $N > $O2/$name.log 2>&1
    done )
  python scripts/pmc_summary.py $O2 > $OUT/r04_pmc_k1_$N.txt 2>&1
done
bash scripts/profile_rgb.sh r04_rgb 2 stats sq1 sq2 fetch wr tcc > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_r04_rgb > gpurun_out/r04_pmc_down512w.txt 2>&1
# structured K2: panel-mark queue (18) and group-mask queue (15) vs register form (12) on config-5-style frame hashes, fetch form (9) on uniform hashes
bash scripts/profile_k2_r04.sh structured18 structured15 structured12 uniform9 > /dev/null 2>&1
echo "profile set done"
