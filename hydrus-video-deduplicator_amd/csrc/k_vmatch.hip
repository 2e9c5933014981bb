// k_vmatch.hip -- the parts of the video-level search (K3, BASELINE config 5) that are not the all-pairs
// kernel itself:
//   * reduction of the device set of (frame, video) keys (hvd_devhash.h) to per-video-pair vPDQ counters
//     (q_hits = distinct frames of a with a match in b, t_hits the converse; vpdqpy/vpdqpy.py:49-56) and their
//     emission as hvd_vmatch records;
//   * list <-> set conversion for the key exchange between ranks (each rank sees only its tiles' hits, a key
//     may be found by two ranks);
//   * VideoHasher.finish() for a whole library on the device: stream compaction of the frames with
//     quality >= tolerance, per-video CSR offsets and the frame -> video map (vpdqpy/vpdqpy.py:119,
//     db/DedupeDB.py:550-553, dedup.py:74-86), so that frames -> hashes -> search never leaves HBM.
// All of this is O(frames) byte shuffling next to the O(frames^2) compare; none of it is on the roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvd_devhash.h"
#include "hvd_kernels.h"

namespace {

using hvd::kEmptyKey;

// ---- set of keys -> pair map ---------------------------------------------------------------------------------
// src: either a table (slots entries, kEmptyKey = free) or a dense list. pkeys/pcnt: open-addressing map
// (a<<32|b) -> (q_hits, t_hits).
__global__ __launch_bounds__(256) void k_keys_to_pairs(const unsigned long long* __restrict__ src, unsigned long long n_src,
                                                       const int32_t* __restrict__ vid_q, const int32_t* __restrict__ vid_t,
                                                       int rect, unsigned long long* __restrict__ pkeys,
                                                       uint2* __restrict__ pcnt, unsigned long long pmask,
                                                       unsigned long long* __restrict__ counters) {
    for (unsigned long long idx = (unsigned long long)blockIdx.x * 256u + threadIdx.x; idx < n_src;
         idx += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long k = src[idx];
        if (k == kEmptyKey) continue;
        const uint32_t f = hvd::vkey_frame(k), v = hvd::vkey_video(k);
        uint32_t a, b;
        bool is_q;
        if (!rect) {  // one frame space: the frame's own video against the video it matched
            const uint32_t own = (uint32_t)vid_q[f];
            is_q = own < v;
            a = is_q ? own : v;
            b = is_q ? v : own;
        } else if (hvd::vkey_side(k) == 0u) {  // query frame matched target video v
            a = (uint32_t)vid_q[f];
            b = v;
            is_q = true;
        } else {  // target frame matched query video v
            a = v;
            b = (uint32_t)vid_t[f];
            is_q = false;
        }
        bool is_new;
        const unsigned long long slot = hvd::table_insert(pkeys, pmask, ((unsigned long long)a << 32) | b, &is_new);
        if (slot == ~0ull) {
            atomicAdd(&counters[0], 1ull);
            continue;
        }
        atomicAdd(is_q ? &pcnt[slot].x : &pcnt[slot].y, 1u);
    }
}

__global__ __launch_bounds__(256) void k_pairs_emit(const unsigned long long* __restrict__ pkeys,
                                                    const uint2* __restrict__ pcnt, unsigned long long slots,
                                                    hvd_vmatch* __restrict__ out, unsigned long long cap,
                                                    unsigned long long* __restrict__ count) {
    for (unsigned long long idx = (unsigned long long)blockIdx.x * 256u + threadIdx.x; idx < slots;
         idx += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long k = pkeys[idx];
        if (k == kEmptyKey) continue;
        const unsigned long long o = atomicAdd(count, 1ull);
        if (o < cap) {
            hvd_vmatch m;
            m.a = (uint32_t)(k >> 32);
            m.b = (uint32_t)k;
            m.q_hits = pcnt[idx].x;
            m.t_hits = pcnt[idx].y;
            out[o] = m;
        }
    }
}

// table -> dense list (for the exchange between ranks); count[0] receives the number of keys
__global__ __launch_bounds__(256) void k_set_to_list(const unsigned long long* __restrict__ tab, unsigned long long slots,
                                                     unsigned long long* __restrict__ list, unsigned long long cap,
                                                     unsigned long long* __restrict__ count) {
    for (unsigned long long idx = (unsigned long long)blockIdx.x * 256u + threadIdx.x; idx < slots;
         idx += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long k = tab[idx];
        if (k == kEmptyKey) continue;
        const unsigned long long o = atomicAdd(count, 1ull);
        if (o < cap) list[o] = k;
    }
}

// dense list (kEmptyKey entries = padding of the all-gather) -> table, de-duplicating
__global__ __launch_bounds__(256) void k_list_to_set(const unsigned long long* __restrict__ list, unsigned long long n,
                                                     unsigned long long* __restrict__ tab, unsigned long long mask,
                                                     unsigned long long* __restrict__ counters) {
    for (unsigned long long idx = (unsigned long long)blockIdx.x * 256u + threadIdx.x; idx < n;
         idx += (unsigned long long)gridDim.x * 256u) {
        const unsigned long long k = list[idx];
        if (k == kEmptyKey) continue;
        bool is_new;
        const unsigned long long slot = hvd::table_insert(tab, mask, k, &is_new);
        if (slot == ~0ull)
            atomicAdd(&counters[0], 1ull);
        else if (is_new)
            atomicAdd(&counters[1], 1ull);
    }
}

// ---- quality filter: stream compaction + CSR ---------------------------------------------------------------
constexpr uint32_t kBlk = 1024;  // frames per workgroup (256 lanes x 4)

__global__ __launch_bounds__(256) void k_keep_count(const int32_t* __restrict__ quality, unsigned long long n, int min_q,
                                                    uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t part[4];
    const unsigned long long base = (unsigned long long)blockIdx.x * kBlk + threadIdx.x * 4u;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n && quality[base + k] >= min_q) ++c;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the block sums in place (one workgroup walks them in chunks of 1024 with a carry);
// total[0] = number of kept frames
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint32_t* __restrict__ sums, uint32_t nb,
                                                          unsigned long long* __restrict__ total) {
    __shared__ uint32_t buf[1024];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nb; c0 += 1024u) {
        const uint32_t idx = c0 + threadIdx.x;
        const uint32_t v = idx < nb ? sums[idx] : 0u;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024u; off <<= 1) {  // Hillis-Steele inclusive scan
            const uint32_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0u;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        const uint32_t carry = carry_s;
        if (idx < nb) sums[idx] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023u) carry_s = carry + buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry_s;
}

__device__ __forceinline__ uint32_t video_of_frame(const long long* __restrict__ offsets, uint32_t V, unsigned long long f) {
    // last v with offsets[v] <= f (empty videos share an offset with their successor and own no frame)
    uint32_t lo = 0, hi = V;  // invariant: offsets[lo] <= f < offsets[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if ((unsigned long long)offsets[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

// pos[f] = number of kept frames before raw frame f; kept frames are copied to their slot together with their
// video index
__global__ __launch_bounds__(256) void k_keep_scatter(const uint4* __restrict__ hashes, const int32_t* __restrict__ quality,
                                                      unsigned long long n, int min_q,
                                                      const uint32_t* __restrict__ block_prefix,
                                                      const long long* __restrict__ offsets, uint32_t V,
                                                      uint4* __restrict__ out_hashes, int32_t* __restrict__ out_video,
                                                      uint32_t* __restrict__ pos) {
    __shared__ uint32_t wave_sum[4];
    const unsigned long long base = (unsigned long long)blockIdx.x * kBlk + threadIdx.x * 4u;
    bool keep[4];
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        keep[k] = base + k < n && quality[base + k] >= min_q;
        c += keep[k] ? 1u : 0u;
    }
    // exclusive scan of c over the 256 lanes: wave scan by shuffles, then the 4 wave totals
    uint32_t incl = c;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += up;
    }
    if (lane == 63u) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = block_prefix[blockIdx.x] + incl - c;
    for (uint32_t w = 0; w < wave; ++w) before += wave_sum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long f = base + k;
        if (f >= n) break;
        pos[f] = before;
        if (keep[k]) {
            out_hashes[(size_t)before * 2u] = hashes[f * 2u];
            out_hashes[(size_t)before * 2u + 1u] = hashes[f * 2u + 1u];
            out_video[before] = (int32_t)video_of_frame(offsets, V, f);
            ++before;
        }
    }
}

__global__ __launch_bounds__(256) void k_keep_offsets(const long long* __restrict__ offsets, uint32_t V, unsigned long long n,
                                                      const uint32_t* __restrict__ pos,
                                                      const unsigned long long* __restrict__ total,
                                                      long long* __restrict__ out_offsets) {
    const uint32_t u = blockIdx.x * 256u + threadIdx.x;
    if (u > V) return;
    const unsigned long long r = (unsigned long long)offsets[u];
    out_offsets[u] = r >= n ? (long long)total[0] : (long long)pos[r];
}

// frame -> video map from CSR offsets (for libraries that arrive as hashes + offsets)
__global__ __launch_bounds__(256) void k_video_of_frames(const long long* __restrict__ offsets, uint32_t V, unsigned long long n,
                                                         int32_t* __restrict__ out_video) {
    const unsigned long long f = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (f < n) out_video[f] = (int32_t)video_of_frame(offsets, V, f);
}

unsigned grid_for(unsigned long long n) {
    unsigned long long b = (n + 255ull) / 256ull;
    if (b < 1) b = 1;
    if (b > 16384ull) b = 16384ull;  // grid-stride kernels
    return (unsigned)b;
}

}  // namespace

namespace hvd {

hipError_t launch_keys_to_pairs(const unsigned long long* d_src, unsigned long long n_src, const int32_t* d_vid_q,
                                const int32_t* d_vid_t, bool rect, unsigned long long* d_pkeys, void* d_pcnt,
                                unsigned long long pmask, unsigned long long* d_counters, hipStream_t s) {
    if (n_src == 0) return hipSuccess;
    hipLaunchKernelGGL(k_keys_to_pairs, dim3(grid_for(n_src)), dim3(256), 0, s, d_src, n_src, d_vid_q, d_vid_t, rect ? 1 : 0,
                       d_pkeys, (uint2*)d_pcnt, pmask, d_counters);
    return hipGetLastError();
}

hipError_t launch_pairs_emit(const unsigned long long* d_pkeys, const void* d_pcnt, unsigned long long slots, hvd_vmatch* d_out,
                             unsigned long long cap, unsigned long long* d_count, hipStream_t s) {
    hipLaunchKernelGGL(k_pairs_emit, dim3(grid_for(slots)), dim3(256), 0, s, d_pkeys, (const uint2*)d_pcnt, slots, d_out, cap,
                       d_count);
    return hipGetLastError();
}

hipError_t launch_set_to_list(const unsigned long long* d_tab, unsigned long long slots, unsigned long long* d_list,
                              unsigned long long cap, unsigned long long* d_count, hipStream_t s) {
    hipLaunchKernelGGL(k_set_to_list, dim3(grid_for(slots)), dim3(256), 0, s, d_tab, slots, d_list, cap, d_count);
    return hipGetLastError();
}

hipError_t launch_list_to_set(const unsigned long long* d_list, unsigned long long n, unsigned long long* d_tab,
                              unsigned long long mask, unsigned long long* d_counters, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_list_to_set, dim3(grid_for(n)), dim3(256), 0, s, d_list, n, d_tab, mask, d_counters);
    return hipGetLastError();
}

size_t compact_scratch_bytes(unsigned long long n) {
    const unsigned long long nb = (n + kBlk - 1) / kBlk;
    return (size_t)(4ull * (nb + 1) + 4ull * (n + 1) + 16ull);
}

// d_scratch: compact_scratch_bytes(n). d_total: one uint64 (device) receiving the kept count.
hipError_t launch_compact_kept(const void* d_hashes, const int32_t* d_quality, unsigned long long n, const long long* d_offsets,
                               uint32_t V, int min_q, void* d_out_hashes, long long* d_out_offsets, int32_t* d_out_video,
                               void* d_scratch, unsigned long long* d_total, hipStream_t s) {
    const unsigned long long nb = (n + kBlk - 1) / kBlk;
    uint32_t* sums = (uint32_t*)d_scratch;
    uint32_t* pos = sums + (nb + 1);
    if (n > 0) {
        hipLaunchKernelGGL(k_keep_count, dim3((unsigned)nb), dim3(256), 0, s, d_quality, n, min_q, sums);
        hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, sums, (uint32_t)nb, d_total);
        hipLaunchKernelGGL(k_keep_scatter, dim3((unsigned)nb), dim3(256), 0, s, (const uint4*)d_hashes, d_quality, n, min_q,
                           sums, d_offsets, V, (uint4*)d_out_hashes, d_out_video, pos);
    } else {
        hipError_t e = hipMemsetAsync(d_total, 0, 8, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_keep_offsets, dim3((V + 1u + 255u) / 256u), dim3(256), 0, s, d_offsets, V, n, pos, d_total,
                       d_out_offsets);
    return hipGetLastError();
}

hipError_t launch_video_of_frames(const long long* d_offsets, uint32_t V, unsigned long long n, int32_t* d_out_video,
                                  hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_video_of_frames, dim3((unsigned)((n + 255ull) / 256ull)), dim3(256), 0, s, d_offsets, V, n,
                       d_out_video);
    return hipGetLastError();
}

}  // namespace hvd
