#!/bin/bash
# Round-2 profile set (run on the GPU box from the repo root): bash scripts/profile_r02.sh
#   1. rocprofv3 --kernel-trace --stats of bench.py itself (same command as the bench line, short legs)
#   2. PMC passes of the dominant kernels (scripts/profile_pmc.sh: counters in their own runs)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT/r02_stats
cd /tmp && export TMPDIR=/tmp
# --no-extras: headline + frames_hashed only, so that the dominant kernel's calls in the summary are exactly the
# warm-up + timed steps of the bench line (the full run adds one 1.8 s launch of the same kernel for config 4)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02_stats -o bench -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $OUT/r02_bench_prof.json 2> $OUT/r02_bench_prof.err
echo "bench under rocprof rc=$?"
find $OUT/r02_stats -name "*kernel_stats.csv" | head -3
cd $REPO && bash scripts/profile_pmc.sh r02
python scripts/pmc_summary.py gpurun_out/pmc_r02 > gpurun_out/r02_pmc_summary.txt 2>&1
tail -5 gpurun_out/r02_pmc_summary.txt
# 3. traffic + SQ counters of the two frame-hash kernels (64x64 gray at 10k and 400k frames; 512x512 rgb at 6144)
cd $REPO
for N in 10000 400000; do
  O2=$OUT/pmc_r02_k1_$N; mkdir -p $O2
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $O2/stats -o stats -- python $REPO/scripts/prof_k1.py $N > $O2/stats.log 2>&1
    for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
      set -- $grp; name=$1; shift
      rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O2/$name -o $name -- python $REPO/scripts/prof_k1.py $N > $O2/$name.log 2>&1
    done )
  python scripts/pmc_summary.py $O2 > $OUT/r02_pmc_k1_$N.txt 2>&1
done
bash scripts/profile_rgb.sh r02_rgb 1 stats sq1 sq2 fetch wr tcc > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_r02_rgb > gpurun_out/r02_pmc_down512w.txt 2>&1
echo "profile set done"
