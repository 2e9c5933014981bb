#!/bin/bash
# The round's profile set (run on the GPU box from the repo root): bash scripts/profile_round.sh r06
#   1. rocprofv3 --kernel-trace --stats of bench.py itself (same command as the bench line, headline + frames legs)
#   2. PMC passes of the uniform all-pairs pass (scripts/profile_pmc.sh: counters in their own runs)
#   3. 64x64 hash kernel at 10k / 400k frames
#   4. the config-5 video search (first stage on the probe's selection -- bits 0..63 + 192..255 --, chosen form alone): durations + HBM-side traffic + MFMA counters
set -u
TAG=${1:-r06}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT/${TAG}_stats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o bench -- python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $OUT/${TAG}_bench_prof.json 2> $OUT/${TAG}_bench_prof.err
echo "bench under rocprof rc=$?"
cp $(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
cd $REPO && bash scripts/profile_pmc.sh ${TAG} > /dev/null 2>&1
python scripts/pmc_summary.py gpurun_out/pmc_${TAG} > gpurun_out/${TAG}_pmc_summary.txt 2>&1
tail -5 gpurun_out/${TAG}_pmc_summary.txt
for N in 10000 400000; do
  O2=$OUT/pmc_${TAG}_k1_$N; mkdir -p $O2
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $O2/stats -o stats -- python $REPO/scripts/prof_k1.py $N > $O2/stats.log 2>&1
    for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
      set -- $grp; name=$1; shift
      rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O2/$name -o $name -- python $REPO/scripts/prof_k1.py $N > $O2/$name.log 2>&1
    done )
  python scripts/pmc_summary.py $O2 > $OUT/${TAG}_pmc_k1_$N.txt 2>&1
done
O5=$OUT/pmc_${TAG}_cfg5; mkdir -p $O5
( cd /tmp && export TMPDIR=/tmp
  CMD="python $REPO/scripts/gpu_cfg5_stages.py"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O5/stats -o stats -- $CMD > $O5/stats.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O5/$c -o $c -- $CMD > $O5/$c.log 2>&1
  done
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d $O5/sq -o sq -- $CMD > $O5/sq.log 2>&1 )
python scripts/pmc_summary.py $O5 > $OUT/${TAG}_pmc_cfg5.txt 2>&1
tail -16 $O5/stats.log > $OUT/${TAG}_cfg5_stages.txt
python scripts/trace_cfg5.py $O5/stats >> $OUT/${TAG}_cfg5_stages.txt 2>&1
echo "profile set done"
