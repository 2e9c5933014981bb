// hvd_devhash.h -- open-addressing tables in HBM for the device-side video-level reduction (K3).
//
// The video-level vPDQ counters (vpdqpy/vpdqpy.py:49-56: share of query frames that have a match in the target
// and the converse) are counts of DISTINCT frames, while the all-pairs kernel meets a frame once per matching
// partner. The reduction therefore needs a set: key = (frame f, video v) = "frame f has at least one frame of
// video v within the tolerance". Two near-duplicate 2-hour videos produce ~5e7 frame-level hits but only 14 400
// distinct keys, which is why the set lives on the device and only video-level records ever leave it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvd {

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr uint32_t kMaxProbe = 512;

// key layout: side(1) | frame(32) | video(31). side 0 = row/query frame, 1 = column/target frame of the
// rectangular form (both frame spaces coincide in the symmetric form, where side is always 0).
__host__ __device__ __forceinline__ unsigned long long vkey_make(uint32_t side, uint32_t frame, uint32_t video) {
    return ((unsigned long long)side << 63) | ((unsigned long long)frame << 31) | (unsigned long long)(video & 0x7FFFFFFFu);
}
__host__ __device__ __forceinline__ uint32_t vkey_side(unsigned long long k) { return (uint32_t)(k >> 63); }
__host__ __device__ __forceinline__ uint32_t vkey_frame(unsigned long long k) { return (uint32_t)(k >> 31); }
__host__ __device__ __forceinline__ uint32_t vkey_video(unsigned long long k) { return (uint32_t)(k & 0x7FFFFFFFull); }

#if defined(__HIPCC__)  // device code below: the host layer also builds with a plain C++ compiler (make asan: g++ + ASan / UBSan)
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// Insert-or-find with linear probing. Returns the slot, or ~0 when the probe limit is hit (the caller counts
// that as a failure and the host re-runs the pass with a larger table). *is_new tells whether this call
// claimed the slot.
__device__ __forceinline__ unsigned long long table_insert(unsigned long long* __restrict__ tab, unsigned long long mask,
                                                           unsigned long long key, bool* is_new) {
    unsigned long long h = mix64(key) & mask;
    *is_new = false;
    for (uint32_t p = 0; p < kMaxProbe; ++p) {
        unsigned long long cur = __atomic_load_n(&tab[h], __ATOMIC_RELAXED);
        if (cur == key) return h;
        if (cur == kEmptyKey) {
            cur = atomicCAS(&tab[h], kEmptyKey, key);
            if (cur == kEmptyKey) {
                *is_new = true;
                return h;
            }
            if (cur == key) return h;
        }
        h = (h + 1) & mask;
    }
    return ~0ull;
}

#endif  // __HIPCC__

// What the all-pairs kernel needs to reduce to video level instead of appending frame pairs.
struct VideoSink {
    unsigned long long* set;       // nullptr => frame-pair mode
    unsigned long long mask;       // slots - 1 (power of two)
    unsigned long long* counters;  // [0] failed inserts, [1] keys newly inserted
    const int32_t* vid_q;          // video index of every row frame
    const int32_t* vid_t;          // video index of every column frame (== vid_q in the symmetric form)
};

#if defined(__HIPCC__)
__device__ __forceinline__ void sink_insert(const VideoSink& vs, unsigned long long key) {
    bool is_new;
    const unsigned long long slot = table_insert(vs.set, vs.mask, key, &is_new);
    if (slot == ~0ull)
        atomicAdd(&vs.counters[0], 1ull);
    else if (is_new)
        atomicAdd(&vs.counters[1], 1ull);
}
#endif  // __HIPCC__

}  // namespace hvd
