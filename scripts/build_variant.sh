#!/bin/bash
# Builds an A/B copy of libhvd_mi355x.so with extra defines for ONE translation unit, for same-box comparisons through
# HVD_LIB_PATH (the ablation / A-B figures under profiles/ were made this way).
#   bash scripts/build_variant.sh nocascade k_hamming_mfma.hip -DHVD_K2_CASCADE=0     -> build_tmp/libhvd_nocascade.so
#   bash scripts/build_variant.sh NOFETCH   k_pdq.hip          -DHVD_ABL_NOFETCH       (timing-only ablation: wrong results)
set -eu
NAME=$1; UNIT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd); SRC=$ROOT/hydrus-video-deduplicator_amd/csrc; OUT=$ROOT/build_tmp; mkdir -p $OUT
make -C $SRC -s
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950"
# (.cpp units go through hipcc too: csrc/Makefile)
( cd $SRC && /opt/rocm/bin/hipcc $FLAGS "$@" -c $UNIT -o $OUT/${UNIT%.*}_$NAME.o )
OBJS=""
for o in hvd_api hvd_search hvd_comm hvd_stream k_hamming k_hamming_mfma k_fp4_image k_vmatch k_synth k_pdq; do
  if [ "$o" = "${UNIT%.*}" ]; then OBJS="$OBJS $OUT/${o}_$NAME.o"; else OBJS="$OBJS $SRC/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $OBJS -shared -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $OUT/libhvd_$NAME.so
echo "$OUT/libhvd_$NAME.so"
