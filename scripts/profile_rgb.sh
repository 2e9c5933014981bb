#!/bin/bash
# rocprofv3 PMC passes for the 512x512 front-end kernels.
# usage: bash scripts/profile_rgb.sh <tag> <pdq_down512_wave: 0|1|2> [pass names...]   (default: every pass)
set -u
TAG=${1:-rgb}; MODE=${2:-1}; shift 2 || true
WANT=${*:-stats sq1 sq2 sq3 fetch tcc tcp1 tcp2 wr}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  case " $WANT " in *" $name "*) ;; *) return;; esac
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o $name -- python $REPO/scripts/prof_rgb.py $MODE > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
case " $WANT " in *" stats "*)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $REPO/scripts/prof_rgb.py $MODE > $OUT/stats.log 2>&1;; esac
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run sq3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run wr WRITE_SIZE
