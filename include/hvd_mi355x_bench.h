/* hvd_mi355x_bench.h -- NOT part of the drop-in boundary.
 *
 * Entry points that exist for this repository's own tests and bench.py only: a workload generator and a test hook. The
 * reference's FFI for this path would bind none of them; they are kept out of include/hvd_mi355x.h so
 * that header is exactly the interface a maintainer binds. A product build may leave them out of the library with
 * -DHVD_NO_BENCH_SYMBOLS. Same conventions as hvd_mi355x.h (int return = HVD_OK or a negative code). */
#ifndef HVD_MI355X_BENCH_H
#define HVD_MI355X_BENCH_H
#include "hvd_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Workload generator (tests / bench.py, BASELINE config 5): writes n_videos * frames_per_video synthetic 64x64 gray
 * frames for videos [v0, v0 + n_videos) to d_frames. Every frame is a pure function of (seed, video, frame index):
 * smooth random field + noise, ~5 % exact constants; d_copy_of (int32 per video of the WHOLE library, indexed by
 * absolute video number, or NULL): videos with copy_of[v] = s >= 0 are video s with +-2 noise per pixel. */
int hvd_dev_synth_video_frames(void* d_frames, int64_t v0, int64_t n_videos, int frames_per_video, uint64_t seed,
                               const void* d_copy_of);

/* Test hook, host only (no device needed): one frame copy through the streaming hasher's helper-thread pool, exactly as
 * hvd_hasher_push does it for a frame of n bytes with `threads` copy threads. tests/test_copy_pool.py hammers it from
 * several threads with different thread counts (the generation race of round 3's pool needed exactly that). */
int hvd_debug_parallel_copy(void* dst, const void* src, size_t n, int threads);

/* Fault injection, tests only: hvd_debug_set("vmatch_fail_rank", r + 1) makes rank r of the next video-level search fail
 * before the key exchange, so that the agreement step (every rank leaves the collective with the same error instead of
 * hanging) can be tested; 0 = off. Like the two entry points above it is absent from -DHVD_NO_BENCH_SYMBOLS builds (the
 * key is then unknown: HVD_ERR_ARG), and it is deliberately NOT in the key list of hvd_mi355x.h. */

#ifdef __cplusplus
}
#endif
#endif /* HVD_MI355X_BENCH_H */
