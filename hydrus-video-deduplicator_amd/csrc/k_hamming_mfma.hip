// k_hamming_mfma.hip -- all-pairs 256-bit Hamming on the gfx950 matrix cores.
//
// Same contract as k_allpairs (k_hamming.hip): all i<j with hamming(db[i],db[j]) <=
// max_dist, bit-identical pair set. The brute-force pair matrix IS a matrix product once
// every hash bit b is written as the number 1-2b in {+1,-1}:
//      dot(a,b) = sum_k a_k*b_k = 256 - 2*hamming(a,b)
// +-1 are exactly representable in FP4 (e2m1: 0x2 = +1.0, 0xA = -1.0), products are +-1
// and the fp32 accumulation of 256 of them is exact, so thresholding the MFMA output at
// dot >= 256 - 2*max_dist is the same predicate as the popcount kernel's, and the distance
// of a hit is recovered as (256 - dot)/2. v_mfma_f32_32x32x64_f8f6f4 (cbsz=blgp=4) does a
// 32x32 block of comparisons over 64 bits in one instruction (32 cycles/SIMD), against
// 16 half-rate VALU ops per comparison for the popcount form.
//
// Data: k_expand_fp4 writes the "FP4 image" of the DB once: 128 B per hash = 8 chunks of
// 16 B (chunk c = bits 32c..32c+31 as 32 nibbles); chunk c of hash n lives in slot
// c ^ ((n>>1)&7) so that a wave's ds_read_b128 of one chunk of 32 consecutive hashes is
// bank-conflict free. An MFMA operand for k-step s is chunk 2s+(lane>>5) of hash
// (lane&31): A and B use the same element order, so the pairing of k indices inside the
// instruction is irrelevant.
//
// Kernel: workgroup = 4 waves; wave w keeps TILES x 32 query hashes as A fragments in
// VGPRs for the whole tile (TILES*16 VGPRs); candidates stream through LDS in
// super-panels of 128 hashes (16 KB, double buffered with direct global->LDS loads, one
// barrier per super-panel). For
// each 32-candidate panel and each query tile: 2 (PREFILTER: first 128 bits) or 4 MFMAs,
// a 16-register max tree, and one wave-uniform branch per panel into the rare path that
// recomputes the full distance and appends the pairs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvd_devhash.h"
#include "hvd_kernels.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mfma_fp4(v4i a, v4i b, v16f c) {
    const v8i xa = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
    const v8i xb = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    // cbsz = blgp = 4: both operands FP4 e2m1; scale 0 -> the unscaled instruction
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, xb, c, 4, 4, 0, 0, 0, 0);
}

__device__ __forceinline__ uint32_t img_slot(uint32_t hash, uint32_t chunk) { return chunk ^ ((hash >> 1) & 7u); }

// One thread per (hash, chunk). Rows >= n (padding up to n_pad) become FP4 zeros.
__global__ __launch_bounds__(256) void k_expand_fp4(const uint32_t* __restrict__ db, uint32_t n, uint32_t n_pad,
                                                    uint4* __restrict__ img, uint32_t code) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint64_t)n_pad * 8u) return;
    const uint32_t hash = (uint32_t)(idx >> 3), chunk = (uint32_t)(idx & 7u);
    uint32_t o[4] = {0u, 0u, 0u, 0u};
    if (hash < n) {
        const uint32_t w = db[(size_t)hash * 8u + chunk];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t x = 0x11111111u * code;  // +v in every nibble (code 2 = +1.0)
#pragma unroll
            for (int t = 0; t < 8; ++t) x |= ((w >> (8 * d + t)) & 1u) << (4 * t + 3);  // bit set -> sign -> -1.0
            o[d] = x;
        }
    }
    img[(size_t)hash * 8u + img_slot(hash, chunk)] = make_uint4(o[0], o[1], o[2], o[3]);
}

// Max of the 16 accumulator registers, taken on the BIT PATTERNS as signed integers:
// for a positive threshold, "float >= thr" and "bits >= thr_bits" agree (negative floats
// have the sign bit set and compare below every positive pattern), and v_max3_i32 needs no
// NaN canonicalisation. (The host routes max_dist >= 128, where thr <= 0, to the popcount
// kernel.)
__device__ __forceinline__ int max16_bits(const v16f& c) {
    int m0 = max(max(__float_as_int(c[0]), __float_as_int(c[1])), __float_as_int(c[2]));
    int m1 = max(max(__float_as_int(c[3]), __float_as_int(c[4])), __float_as_int(c[5]));
    int m2 = max(max(__float_as_int(c[6]), __float_as_int(c[7])), __float_as_int(c[8]));
    int m3 = max(max(__float_as_int(c[9]), __float_as_int(c[10])), __float_as_int(c[11]));
    int m4 = max(max(__float_as_int(c[12]), __float_as_int(c[13])), __float_as_int(c[14]));
    m0 = max(max(m0, m1), m2);
    m3 = max(max(m3, m4), __float_as_int(c[15]));
    return max(m0, m3);
}

__device__ __forceinline__ void append_pair_m(hvd_pair* out, unsigned long long cap, unsigned long long* count,
                                              uint32_t i, uint32_t j, uint32_t dist) {
    unsigned long long slot = atomicAdd(count, 1ull);
    if (slot < cap) {
        hvd_pair p;
        p.i = i;
        p.j = j;
        p.dist = dist;
        p.pad = 0;
        out[slot] = p;
    }
}

constexpr int kSuper = 128;  // candidates per LDS super-panel (256: -4 % with the prefilter, +2 % without)

__device__ __forceinline__ v4i as_v4i(const uint4& v) { return v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }

template <bool PREFILTER>
__device__ __forceinline__ v16f tile_dot(const v4i (&a)[4], const v4i (&b)[4]) {
    const v16f z = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    v16f acc = mfma_fp4(a[0], b[0], z);
    acc = mfma_fp4(a[1], b[1], acc);
    if (!PREFILTER) {
        acc = mfma_fp4(a[2], b[2], acc);
        acc = mfma_fp4(a[3], b[3], acc);
    }
    return acc;
}

// Rare path: some pair of this 32-candidate panel may be within tolerance. Recompute every
// query tile of the wave over all 256 bits (A fragments are re-read from memory so that the
// loop stays rolled and the fast path's registers stay untouched) and report the hits.
// rect = false: one set, pairs i<j, group[i] != group[j]. rect = true: query set x target set
// (row index into the query image, column index into the target image), every (i<nq, j<n) pair.
//
// Two sinks. Frame-pair mode (vs.set == nullptr): one hvd_pair per hit. Video mode (K3): a hit (i, j) means
// "frame i has a match in video(j)" and "frame j has a match in video(i)"; those two facts go into the
// device set as keys, de-duplicated inside the panel before any atomic is issued: frames are stored in video
// order, so the 32 columns of a panel and the rows a lane walks both visit videos monotonically -- a row key is
// issued only by the first hit column of its video (ballot of the row's hits against the panel's video
// segments), a column key only when the row video changes. A panel in which every pair matches (two copies of
// one video) costs 2 atomics per row and column instead of 1024 appends.
template <int TILES>
__device__ __noinline__ void panel_slow_path(const uint4* __restrict__ img, const uint4* base, uint32_t sw,
                                             uint32_t wrow0, uint32_t j, uint32_t n, uint32_t h, uint32_t li,
                                             const int32_t* __restrict__ group, float thr_full,
                                             hvd_pair* __restrict__ out, unsigned long long cap,
                                             unsigned long long* __restrict__ count, bool rect, uint32_t nq,
                                             const int32_t* __restrict__ group_t, float inv_scale2,
                                             const hvd::VideoSink vs) {
    v4i bf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bf[s] = as_v4i(base[(2u * s + h) ^ sw]);
    const bool video = vs.set != nullptr;
    const int32_t gcol = (group != nullptr && j < n) ? (rect ? group_t[j] : group[j]) : 0;
    int32_t vcol = -1, last_v = -1;
    uint32_t lowmask = 0;
    if (video) {
        vcol = j < n ? vs.vid_t[j] : -1;
        const int32_t vprev = __shfl_up(vcol, 1);
        const unsigned long long seg = __ballot(li == 0u || vcol != vprev);  // first column of each video in the panel
        const uint32_t segh = h ? (uint32_t)(seg >> 32) : (uint32_t)seg;
        const uint32_t upto = segh & (0xFFFFFFFFu >> (31u - li));  // bit 0 is always set
        const uint32_t segstart = 31u - (uint32_t)__clz((int)upto);
        lowmask = ((1u << li) - 1u) & ~((1u << segstart) - 1u);  // lower columns of my video
    }
#pragma unroll 1
    for (int t = 0; t < TILES; ++t) {
        const uint32_t hash = wrow0 + 32u * t + li;
        v4i af[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) af[s] = as_v4i(img[(size_t)hash * 8u + img_slot(hash, 2u * s + h)]);
        const v16f acc = tile_dot<false>(af, bf);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
            const uint32_t i = wrow0 + 32u * t + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * h;
            bool ok = acc[r] >= thr_full && j < n && (rect ? i < nq : i < j);
            if (ok && group != nullptr) ok = group[i] != gcol;
            if (!video) {
                if (ok) append_pair_m(out, cap, count, i, j, (uint32_t)(256 - (int)(acc[r] * inv_scale2)) >> 1);
                continue;
            }
            const unsigned long long rowhits = __ballot(ok);
            if (rowhits == 0ull) continue;  // wave-uniform
            const uint32_t mine = h ? (uint32_t)(rowhits >> 32) : (uint32_t)rowhits;
            if (ok) {
                if ((mine & lowmask) == 0u) hvd::sink_insert(vs, hvd::vkey_make(0u, i, (uint32_t)vcol));
                const int32_t vrow = vs.vid_q[i];
                if (vrow != last_v) {
                    hvd::sink_insert(vs, hvd::vkey_make(rect ? 1u : 0u, j, (uint32_t)vrow));
                    last_v = vrow;
                }
            }
        }
    }
}

// Stage one 16 KB super-panel (128 hashes x 128 B, contiguous in the image) into LDS with
// direct global->LDS loads: each wave-instruction moves 64 lanes x 16 B = 1 KB to a
// wave-uniform LDS base + lane*16, no VGPR round trip (so nothing to keep live -- or spill --
// across the compute phase). The data is complete after the vmcnt(0) that hipcc places in
// front of the next __syncthreads().
__device__ __forceinline__ void stage_super_panel(const uint4* __restrict__ src, uint4* lds_dst, uint32_t wave,
                                                  uint32_t lane) {
#pragma unroll
    for (int q = 0; q < kSuper * 8 / 256; ++q) {
        const uint32_t chunk0 = (uint32_t)q * 256u + wave * 64u;  // first 16-B chunk of this wave-instruction
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + chunk0 + lane),
            (__attribute__((address_space(3))) void*)(lds_dst + chunk0), 16, 0, 0);
    }
}

// RECT = false: img_q == img (one set, strict upper triangle). RECT = true: rows come from the
// query image img_q (nq hashes), candidates from the target image img (n hashes), full rectangle.
template <int TILES, bool PREFILTER, bool RECT>
__global__ __launch_bounds__(256, 2) void k_allpairs_mfma(const uint4* __restrict__ img, uint32_t n, uint32_t n_pad,
                                                          const int32_t* __restrict__ group, uint32_t max_dist,
                                                          uint32_t col_chunk, uint32_t rank, uint32_t world,
                                                          hvd_pair* __restrict__ out, unsigned long long cap,
                                                          unsigned long long* __restrict__ count,
                                                          const uint4* __restrict__ img_q, uint32_t nq,
                                                          const int32_t* __restrict__ group_t, float scale2,
                                                          const hvd::VideoSink vs) {
    constexpr uint32_t WROWS = 32u * TILES, ROWS = 4u * WROWS;
    constexpr int NB = PREFILTER ? 2 : 4;
    __shared__ uint4 lds0[kSuper * 8], lds1[kSuper * 8];

    const uint32_t rb = blockIdx.x, cb = blockIdx.y;
    const uint32_t row0 = rb * ROWS;
    const uint32_t col0 = cb * col_chunk;
    const uint32_t col1 = min(col0 + col_chunk, n_pad);
    if (!RECT && min(col1, n) <= row0 + 1u) return;  // tile entirely on/below the diagonal
    if (world > 1u && (rb + cb) % world != rank) return;
    const uint4* __restrict__ imgq = RECT ? img_q : img;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, li = lane & 31u, h = lane >> 5;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wrow0 = row0 + wave * WROWS;

    // A fragments: query hash (wrow0 + 32t + li), chunk 2s+h, for the whole tile.
    v4i a[TILES][4];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const uint32_t hash = wrow0 + 32u * t + li;  // < n_pad by construction
#pragma unroll
        for (int s = 0; s < NB; ++s) a[t][s] = as_v4i(imgq[(size_t)hash * 8u + img_slot(hash, 2u * s + h)]);
    }

    // every product is +-v*v = +-scale2, so all dot products (and thresholds) scale by scale2
    const float thr_full = scale2 * (256.0f - 2.0f * (float)max_dist);                       // > 0 (host guarantees)
    const float thr_fast = PREFILTER ? scale2 * (128.0f - 2.0f * (float)max_dist) : thr_full;  // may be <= 0: then every
    const int thr_bits = thr_fast > 0.0f ? __float_as_int(thr_fast) : (int)0x80000000;  // panel takes the slow path

    // candidates <= row0 cannot pair with rows >= row0 (i<j): start at the super-panel holding row0+1
    const uint32_t j0 = RECT ? col0 : max(col0, (row0 + 1u) & ~(uint32_t)(kSuper - 1));
    const uint32_t nsp = (col1 - j0) / kSuper;  // col0, col1, j0 are multiples of kSuper

    // The two LDS buffers are separate objects and the super-panel loop is unrolled by two, so that every
    // ds_read names one array and every in-flight global->LDS load the other: with one two-dimensional array
    // the compiler could not tell them apart and put s_waitcnt vmcnt(0) in front of every panel's LDS reads,
    // i.e. it waited for the prefetch of the NEXT super-panel before computing on this one.
    auto process = [&](const uint4* __restrict__ panel, const uint32_t jsp) {
        // (reading the next panel's B fragments one panel early was measured twice: +27 VGPRs and 3-6 % slower)
#pragma unroll 1
        for (uint32_t p = 0; p < kSuper / 32; ++p) {
            const uint32_t cl = 32u * p + li;  // candidate index inside the super-panel
            const uint4* base = &panel[cl * 8u];
            const uint32_t sw = (cl >> 1) & 7u;  // jsp is a multiple of 128: same swizzle as the global index
            v4i b[4];
#pragma unroll
            for (int s = 0; s < NB; ++s) b[s] = as_v4i(base[(2u * s + h) ^ sw]);

            // two accumulator sets: the MFMAs of tile t+1 run under the max tree of tile t
            int mm = (int)0x80000000;
            v16f cur = tile_dot<PREFILTER>(a[0], b);
#pragma unroll
            for (int t = 1; t < TILES; ++t) {
                const v16f nxt = tile_dot<PREFILTER>(a[t], b);
                mm = max(mm, max16_bits(cur));  // runs in the shadow of tile t's MFMAs (other accumulator set)
                cur = nxt;
            }
            mm = max(mm, max16_bits(cur));

            if (__builtin_expect(__any(mm >= thr_bits), 0))
                panel_slow_path<TILES>(imgq, base, sw, wrow0, jsp + cl, n, h, li, group, thr_full, out, cap, count,
                                       RECT, nq, group_t, 1.0f / scale2, vs);
        }
    };

    stage_super_panel(img + (size_t)j0 * 8u, lds0, wave, lane);
    __syncthreads();

    for (uint32_t sp = 0; sp < nsp; sp += 2) {
        const uint32_t jsp = j0 + sp * kSuper;
        // lds1 was last read in iteration sp-1, which every wave left through a barrier
        if (sp + 1u < nsp) stage_super_panel(img + (size_t)(jsp + kSuper) * 8u, lds1, wave, lane);
        process(lds0, jsp);
        __syncthreads();  // (drains the in-flight global->LDS loads with vmcnt(0) first)
        if (sp + 1u >= nsp) break;
        if (sp + 2u < nsp) stage_super_panel(img + (size_t)(jsp + 2u * kSuper) * 8u, lds0, wave, lane);
        process(lds1, jsp + kSuper);
        __syncthreads();
    }
}

}  // namespace

namespace hvd {

static uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

uint32_t fp4_rows_padded(uint32_t n) { return round_up(n ? n : 1, 1024u); }

// e2m1 code of the magnitude used for the +-v image: 1 = 0.5, 2 = 1.0 (default), 4 = 2.0, 6 = 4.0.
// Any of them is exact; the choice only changes what toggles in the multiplier array (power -> clock).
uint32_t g_fp4_code = 2;
static float fp4_scale2() {
    const float v = g_fp4_code == 1 ? 0.5f : g_fp4_code == 2 ? 1.0f : g_fp4_code == 4 ? 2.0f : 4.0f;
    return v * v;
}

hipError_t launch_expand_fp4(const void* d_db, uint32_t n, void* d_img, hipStream_t s) {
    const uint32_t n_pad = fp4_rows_padded(n);
    const uint64_t threads = (uint64_t)n_pad * 8u;
    hipLaunchKernelGGL(k_expand_fp4, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const uint32_t*)d_db, n,
                       n_pad, (uint4*)d_img, g_fp4_code);
    return hipGetLastError();
}

uint32_t g_mfma_col_chunk_max = 8192;  // tuning knob (hvd_debug_set "mfma_col_chunk_max"); 2048..32768 within 3 %

static uint32_t pick_col_chunk_m(uint32_t n_pad, uint32_t rows_per_wg) {
    uint64_t n_rb = (n_pad + rows_per_wg - 1) / rows_per_wg;
    uint64_t want_cb = (8192 + n_rb - 1) / n_rb;
    if (want_cb < 1) want_cb = 1;
    uint64_t chunk = (n_pad + want_cb - 1) / want_cb;
    if (chunk < 256) chunk = 256;
    if (chunk > g_mfma_col_chunk_max) chunk = g_mfma_col_chunk_max;
    chunk = (chunk + kSuper - 1) / kSuper * kSuper;
    if ((n_pad + chunk - 1) / chunk > 65535u) chunk = round_up((n_pad + 65534u) / 65535u, kSuper);
    return (uint32_t)chunk;
}

static int mfma_tiles(int variant) {
    switch (variant) {
        case 8: case 9: return 8;
        case 10: case 11: return 4;
        default: return 0;
    }
}

bool allpairs_mfma_geometry(uint32_t n, int variant, uint32_t* rows_per_block, uint32_t* col_chunk) {
    const int T = mfma_tiles(variant);
    if (!T) return false;
    *rows_per_block = 128u * (uint32_t)T;
    *col_chunk = pick_col_chunk_m(fp4_rows_padded(n), *rows_per_block);
    return true;
}

template <int T, bool PF>
static hipError_t launch_mfma_t(const AllPairsArgs& a, const void* d_img, hipStream_t s) {
    const uint32_t n_pad = fp4_rows_padded(a.n);
    constexpr uint32_t ROWS = 128u * T;
    const uint32_t chunk = pick_col_chunk_m(n_pad, ROWS);
    dim3 grid((a.n + ROWS - 1) / ROWS, (n_pad + chunk - 1) / chunk);
    hipLaunchKernelGGL((k_allpairs_mfma<T, PF, false>), grid, dim3(256), 0, s, (const uint4*)d_img, a.n, n_pad,
                       a.d_group, a.max_dist, chunk, a.rank, a.world, a.d_pairs, a.cap, a.d_count,
                       (const uint4*)nullptr, 0u, (const int32_t*)nullptr, fp4_scale2(), a.sink);
    return hipGetLastError();
}

// Query set (nq hashes, image d_img_q) x target set (a.n hashes, image d_img_t): full rectangle.
template <int T, bool PF>
static hipError_t launch_cross_t(const AllPairsArgs& a, const void* d_img_q, uint32_t nq, const void* d_img_t,
                                 const int32_t* d_group_t, hipStream_t s) {
    const uint32_t n_pad = fp4_rows_padded(a.n);
    constexpr uint32_t ROWS = 128u * T;
    // column chunk sized for the rectangle: enough tiles to fill the chip even when nq is small
    uint64_t n_rb = (nq + ROWS - 1) / ROWS;
    uint64_t want_cb = (4096 + n_rb - 1) / n_rb;
    uint64_t chunk = (n_pad + want_cb - 1) / want_cb;
    if (chunk < 256) chunk = 256;
    if (chunk > 4096) chunk = 4096;
    chunk = (chunk + kSuper - 1) / kSuper * kSuper;
    if ((n_pad + chunk - 1) / chunk > 65535u) chunk = round_up((n_pad + 65534u) / 65535u, kSuper);
    dim3 grid((unsigned)n_rb, (unsigned)((n_pad + chunk - 1) / chunk));
    hipLaunchKernelGGL((k_allpairs_mfma<T, PF, true>), grid, dim3(256), 0, s, (const uint4*)d_img_t, a.n, n_pad,
                       a.d_group, a.max_dist, (uint32_t)chunk, a.rank, a.world, a.d_pairs, a.cap, a.d_count,
                       (const uint4*)d_img_q, nq, d_group_t, fp4_scale2(), a.sink);
    return hipGetLastError();
}

hipError_t launch_cross_mfma(const AllPairsArgs& a, const void* d_img_q, uint32_t nq, const void* d_img_t,
                             const int32_t* d_group_t, hipStream_t s) {
    if (nq == 0 || a.n == 0) return hipSuccess;
    if (a.max_dist >= 128u) return hipErrorInvalidValue;  // sign trick needs a positive threshold
    if (a.max_dist >= 64u) return launch_cross_t<8, false>(a, d_img_q, nq, d_img_t, d_group_t, s);
    return launch_cross_t<8, true>(a, d_img_q, nq, d_img_t, d_group_t, s);
}

hipError_t launch_allpairs_mfma(const AllPairsArgs& a_in, const void* d_img, hipStream_t s) {
    if (a_in.n < 2) return hipSuccess;
    AllPairsArgs a = a_in;
    // The sign trick of max16_bits needs a positive threshold: 256 - 2*max_dist > 0. Larger
    // tolerances (never used by the reference, which fixes 31) go to the popcount kernel, which
    // shares the tile geometry class; the 128-bit prefilter needs 128 - 2*max_dist > 0.
    if (a.max_dist >= 128u) {
        if (!a.d_db) return hipErrorInvalidValue;
        a.variant = 0;
        return launch_allpairs(a, s);
    }
    if (a.max_dist >= 64u && (a.variant == 9 || a.variant == 11)) a.variant -= 1;
    switch (a.variant) {
        case 8: return launch_mfma_t<8, false>(a, d_img, s);
        case 9: return launch_mfma_t<8, true>(a, d_img, s);
        case 10: return launch_mfma_t<4, false>(a, d_img, s);
        case 11: return launch_mfma_t<4, true>(a, d_img, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace hvd
