#!/bin/bash
# The GPU suite's host-heavy tests against the ASan + UBSan build of the host layer (make -C hydrus-video-deduplicator_amd/csrc asan).
# usage (GPU box, repo root): bash scripts/run_asan.sh [pytest -k expression]   -> gpurun_out/asan/{pytest.log,asan.*}
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/asan; mkdir -p $OUT
LIB=$REPO/hydrus-video-deduplicator_amd/libhvd_mi355x_asan.so
[ -f $LIB ] || make -C $REPO/hydrus-video-deduplicator_amd/csrc asan || exit 1
RT=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)
K=${1:-"group or hasher or match_server or k3 or vmatch or abort or stream or cross or sqlite"}
# detect_leaks=0: python itself leaks by design;
# log_path: one file per process (the group tests run in child processes)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:log_path=$OUT/asan:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1:log_path=$OUT/ubsan
HVD_LIB_PATH=$LIB LD_PRELOAD=$RT python -m pytest tests -m gpu -q -p no:cacheprovider -k "$K" > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
ls $OUT | head -20
cat $OUT/asan.* $OUT/ubsan.* 2>/dev/null | grep -E "ERROR: AddressSanitizer|runtime error|SUMMARY" | sort | uniq -c | head -40
