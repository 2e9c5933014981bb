#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration for the access shapes of k_down512w (scripts/ubench_fetch_calib.hip), with the real
# kernel's own fetch / write passes taken on the same box. usage (GPU box, repo root): bash scripts/profile_fetch_calib.sh <tag>
set -u
TAG=${1:-calib}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
BIN=$REPO/scripts/ubench_fetch_calib
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/scripts/ubench_fetch_calib.hip -o $BIN || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*" | sort -u > $OUT/avail.txt
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- $BIN > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BIN > $OUT/stats.log 2>&1; echo "stats rc=$?"
run fetch FETCH_SIZE
run wr WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
# the real kernel, same box
rk() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/k_$name -o $name -- python $REPO/scripts/prof_rgb.py 2 > $OUT/k_$name.log 2>&1; echo "k_$name rc=$?"; }
rk fetch FETCH_SIZE
rk wr WRITE_SIZE
rk rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
rk wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python $REPO/scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -150
