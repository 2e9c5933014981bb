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
// Data (k_fp4_image.hip, hvd_fp4.h): k_expand_fp4 writes the "FP4 image" of the DB once: 128 B per hash = 8 chunks of
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
#include <string.h>

#include <mutex>

#include "hvd_devhash.h"
#include "hvd_fp4.h"
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

// The fast path's form of the same test. The query fragments are NEGATED (+-1 -> -+1: one XOR of the sign nibbles at
// load) and the accumulator starts at (threshold - 1): acc = (thr - 1) - dot. Dot products of +-1 vectors of even
// length are even, so acc is an odd multiple of the scale -- never 0 -- and NEGATIVE exactly when dot >= thr: a hit is
// a set sign bit, and "any of these 16" is the sign of their bitwise OR. v_or3_b32 issues at the full VALU rate where
// v_max3_i32 takes two slots (profiles/r01_ubench_valu.txt), and the epilogue is what the power-limited clock pays for.
__device__ __forceinline__ int or16_bits(const v16f& c) {
    int m0 = __float_as_int(c[0]) | __float_as_int(c[1]) | __float_as_int(c[2]);
    int m1 = __float_as_int(c[3]) | __float_as_int(c[4]) | __float_as_int(c[5]);
    int m2 = __float_as_int(c[6]) | __float_as_int(c[7]) | __float_as_int(c[8]);
    int m3 = __float_as_int(c[9]) | __float_as_int(c[10]) | __float_as_int(c[11]);
    int m4 = __float_as_int(c[12]) | __float_as_int(c[13]) | __float_as_int(c[14]);
    m0 = m0 | m1 | m2;
    m3 = m3 | m4 | __float_as_int(c[15]);
    return m0 | m3;
}


__device__ __forceinline__ float min16(const v16f& c) {
    float m0 = fminf(fminf(c[0], c[1]), c[2]);
    float m1 = fminf(fminf(c[3], c[4]), c[5]);
    float m2 = fminf(fminf(c[6], c[7]), c[8]);
    float m3 = fminf(fminf(c[9], c[10]), c[11]);
    float m4 = fminf(fminf(c[12], c[13]), c[14]);
    m0 = fminf(fminf(m0, m1), m2);
    m3 = fminf(fminf(m3, m4), c[15]);
    return fminf(m0, m3);
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

// Frame-pair sink, round 3: hits are collected per WORKGROUP in LDS and flushed with ONE global atomic when the
// workgroup is done. `count` is a single address shared by the whole chip, and device-scope atomics on one address
// serialise at the memory side (round 2 measured ~8 ns per appended pair on a clustered DB: 41 ms for 4.95 M pairs --
// the append, not the compare, set the pace). A workgroup walks 8.4 M comparisons; even a heavily clustered DB leaves
// well under kWgPairs hits per workgroup, and the overflow (one workgroup covering two copies of the same long video in
// frame-pair mode) falls back to the direct append, so nothing is ever dropped. File-scope LDS so that the noinline hit
// handler reaches it without another argument register.
constexpr uint32_t kWgPairs = 512;
__shared__ hvd_pair g_wg_pairs[kWgPairs];
__shared__ uint32_t g_wg_npairs;
__shared__ unsigned long long g_clk_t0[2];  // clock telemetry: a sampled workgroup's start values (see the kernel)

__device__ __forceinline__ void append_pair_wg(hvd_pair* out, unsigned long long cap, unsigned long long* count,
                                               uint32_t i, uint32_t j, uint32_t dist) {
    const uint32_t slot = atomicAdd(&g_wg_npairs, 1u);
    if (slot < kWgPairs) {
        hvd_pair p;
        p.i = i;
        p.j = j;
        p.dist = dist;
        p.pad = 0;
        g_wg_pairs[slot] = p;
    } else {
        append_pair_m(out, cap, count, i, j, dist);
    }
}

constexpr int kSuper = 128;  // candidates per LDS super-panel (256: -4 % with the prefilter, +2 % without)

__device__ __forceinline__ v4i as_v4i(const uint4& v) { return v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w}; }

// k-steps [S0, S1) of a 32x32 tile: 64 hash bits per step.
template <int S0, int S1>
__device__ __forceinline__ v16f tile_dot(const v4i* a, const v4i* b, v16f acc) {
#pragma unroll
    for (int s = S0; s < S1; ++s) acc = mfma_fp4(a[s], b[s], acc);
    return acc;
}

// Everything the hit handler needs that is uniform over the launch (one kernel argument).
struct HitCtx {
    const int32_t* group;    // rows (nullable)
    const int32_t* group_t;  // columns of the rectangular form
    hvd_pair* out;
    unsigned long long cap;
    unsigned long long* count;
    hvd::VideoSink vs;
    uint32_t n, nq;
    float thr_full;  // 256 - 2 * max_dist: a dot product at or above it is a hit
    uint32_t rect;
    float acc_start;  // start value of the launched form's accumulators (see or16_bits)
    const uint4* img_q;  // FP4 images of the rows / of the columns (the pair queue's drain reads single hashes from them)
    const uint4* img_t;
    const uint4* db_q;  // the same hashes packed (32 B each), nullable: the drain's cheaper source (2 loads per hash instead of 8)
    const uint4* db_t;
    uint32_t max_dist;
    unsigned long long* clk;  // clock telemetry: {shader cycles, constant-rate ticks, sampled workgroups} accumulated over passes
};

// Hits of one 32x32 tile whose accumulators hold the full 256-bit dot products: acc[r] belongs to
// row = row0 + (r&3) + 8*(r>>2) + 4*(lane>>5), column j (C/D layout of the 32x32 MFMA).
// rect = 0: one set, pairs i<j, group[i] != group[j]. rect = 1: query set x target set (row index into the
// query image, column index into the target image), every (i<nq, j<n) pair.
//
// Two sinks. Frame-pair mode (vs.set == nullptr): one hvd_pair per hit. Video mode (K3): a hit (i, j) means
// "frame i has a match in video(j)" and "frame j has a match in video(i)"; those two facts go into the
// device set as keys, de-duplicated inside the tile before any atomic is issued: frames are stored in video
// order, so the 32 columns of a panel and the rows a lane walks both visit videos monotonically -- a row key is
// issued only by the first hit column of its video (ballot of the row's hits against the panel's video
// segments), a column key only when the row video changes. A tile in which every pair matches (two copies of
// one video) costs ~2 atomics per row and column instead of 1024 appends.
// A pointer argument of a non-kernel function arrives in VGPRs; make it provably uniform and constant so that the
// context is fetched with scalar loads into SGPRs (as flat loads it occupied ~30 VGPRs, which add to the fast path's
// register footprint: the handlers' registers and the values their caller keeps live across the call must coexist).
// (readfirstlane returns a SIGNED int: widen through uint32_t, or a low dword with bit 31 set smears into the high one)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | (unsigned long long)lo;
#else
    return v;
#endif
}

__device__ __forceinline__ HitCtx load_ctx(const HitCtx* ctx) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(4))) HitCtx*)uniform_u64((unsigned long long)ctx);
#else
    return *ctx;
#endif
}

// End of the workgroup's tile (all waves past their last barrier): reserve room for the collected pairs with one
// global atomic and copy them out. noinline: nothing of the fast path is live any more when it is called, and kept
// out of line it does not disturb the register allocation of the loop (inlined it cost the fetch form 6 VGPRs = one
// resident wave per SIMD).
// (tid is an argument: a callee that reads threadIdx makes the caller keep the packed work-item ids alive in v31.)
__device__ __noinline__ void flush_pairs_wg(const HitCtx* __restrict__ ctx, uint32_t tid, uint32_t nthreads) {
    __shared__ unsigned long long base_s;
    const HitCtx c = load_ctx(ctx);
    if (tid == 0u && g_clk_t0[0] != 0ull) {  // clock telemetry of a sampled workgroup (see the kernel): its lifetime ends here
        atomicAdd(&c.clk[0], (unsigned long long)__builtin_readcyclecounter() - g_clk_t0[0]);
        atomicAdd(&c.clk[1], (unsigned long long)__builtin_amdgcn_s_memrealtime() - g_clk_t0[1]);
        atomicAdd(&c.clk[2], 1ull);
    }
    const uint32_t m = min(g_wg_npairs, kWgPairs);
    if (m == 0u) return;  // uniform over the workgroup
    if (tid == 0u) base_s = atomicAdd(c.count, (unsigned long long)m);
    __syncthreads();
    const unsigned long long base = base_s;
    for (uint32_t k = tid; k < m; k += nthreads)
        if (base + k < c.cap) c.out[base + k] = g_wg_pairs[k];
}

// acc holds sgn * dot + off (sgn = +1, off = 0 for a plain recomputation; sgn = -1, off = start value for the fast
// path's negated accumulators): dot = sgn * (acc - off) ... written as one exact fma on small integers.
__device__ __forceinline__ void tile_hits_body(const v16f& acc, uint32_t row0, uint32_t j, uint32_t lane, const HitCtx& c,
                                               float sgn, float off) {
    const uint32_t li = lane & 31u, h = lane >> 5;
    const bool video = c.vs.set != nullptr;
    const bool rect = c.rect != 0u;
    const int32_t gcol = (c.group != nullptr && j < c.n) ? (rect ? c.group_t[j] : c.group[j]) : 0;
    int32_t vcol = -1, last_v = -1;
    uint32_t lowmask = 0;
    if (video) {
        vcol = j < c.n ? c.vs.vid_t[j] : -1;
        const int32_t vprev = __shfl_up(vcol, 1);
        const unsigned long long seg = __ballot(li == 0u || vcol != vprev);  // first column of each video in the panel
        const uint32_t segh = h ? (uint32_t)(seg >> 32) : (uint32_t)seg;
        const uint32_t upto = segh & (0xFFFFFFFFu >> (31u - li));  // bit 0 is always set
        const uint32_t segstart = 31u - (uint32_t)__clz((int)upto);
        lowmask = ((1u << li) - 1u) & ~((1u << segstart) - 1u);  // lower columns of my video
    }
    // rolled on purpose: the handler's register footprint adds to the fast path's (its caller keeps ~110 values
    // live across the call), and 16 unrolled iterations cost 40 more VGPRs = one resident wave per SIMD
#pragma unroll 1
    for (int r = 0; r < 16; ++r) {
        const uint32_t i = row0 + (uint32_t)((r & 3) + 8 * (r >> 2)) + 4u * h;
        const float dot = fmaf(sgn, acc[r], -sgn * off);
        bool ok = dot >= c.thr_full && j < c.n && (rect ? i < c.nq : i < j);
        if (ok && c.group != nullptr) ok = c.group[i] != gcol;
        if (!video) {
            if (ok) append_pair_wg(c.out, c.cap, c.count, i, j, (uint32_t)(256 - (int)dot) >> 1);
            continue;
        }
        const unsigned long long rowhits = __ballot(ok);
        if (rowhits == 0ull) continue;  // wave-uniform
        const uint32_t mine = h ? (uint32_t)(rowhits >> 32) : (uint32_t)rowhits;
        if (ok) {
            if ((mine & lowmask) == 0u) hvd::sink_insert(c.vs, hvd::vkey_make(0u, i, (uint32_t)vcol));
            const int32_t vrow = c.vs.vid_q[i];
            if (vrow != last_v) {
                hvd::sink_insert(c.vs, hvd::vkey_make(rect ? 1u : 0u, j, (uint32_t)vrow));
                last_v = vrow;
            }
        }
    }
}

// NEG: the accumulators are the fast path's negated ones (start value c.acc_start); otherwise plain dot products.
template <bool NEG>
__device__ __noinline__ void tile_hits(const v16f acc, uint32_t row0, uint32_t j, uint32_t lane,
                                       const HitCtx* __restrict__ ctx) {
    const HitCtx c = load_ctx(ctx);
    tile_hits_body(acc, row0, j, lane, c, NEG ? -1.0f : 1.0f, NEG ? c.acc_start : 0.0f);
}

// Deferred form of the same for the kernels whose query fragments hold only the first 128 bits: `marks` has, per lane,
// one bit per query tile of the wave whose first stage found a candidate in this 32-candidate panel. Those tiles are
// recomputed over all 256 bits, one k-step at a time straight from memory (few registers: see load_ctx), and their
// hits reported.
template <int TILES>
__device__ __noinline__ void panel_survivors(uint32_t marks, const uint4* __restrict__ imgq_v, const uint4* panel_v,
                                             uint32_t wrow0_v, uint32_t j0_v, uint32_t lane, const HitCtx* __restrict__ ctx) {
    // Everything but the lane id is wave-uniform; arguments of a non-kernel function arrive in VGPRs, so move them to
    // SGPRs: what stays live across the nested handler call below adds to the fast path's register footprint.
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t wrow0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wrow0_v);
    const uint32_t j0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)j0_v);
#else
    const uint32_t wrow0 = wrow0_v, j0 = j0_v;
#endif
    const uint4* __restrict__ imgq = (const uint4*)uniform_u64((unsigned long long)imgq_v);
    const uint4* panel = (const uint4*)uniform_u64((unsigned long long)panel_v);
    const int thr2_bits = __float_as_int(load_ctx(ctx).thr_full);
#pragma unroll 1
    for (int t = 0; t < TILES; ++t) {
        if (!__any((marks >> (TILES - 1 - t)) & 1u)) continue;  // (per-lane marks: bit TILES-1-t <-> tile t)
        const uint32_t li = lane & 31u, h = lane >> 5;
        const uint32_t cl = (j0 & (uint32_t)(kSuper - 1)) + li;  // candidate index inside the super-panel
        const uint4* base = &panel[cl * 8u];
        const uint32_t sw = (cl >> 1) & 7u;
        const uint32_t hash = wrow0 + 32u * (uint32_t)t + li;
        v16f acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (uint32_t s_ = 0; s_ < 4u; ++s_) {
            const v4i af = as_v4i(imgq[(size_t)hash * 8u + img_slot(hash, 2u * s_ + h)]);
            const v4i bf = as_v4i(base[(2u * s_ + h) ^ sw]);
            acc = mfma_fp4(af, bf, acc);
        }
        if (!__any(max16_bits(acc) >= thr2_bits)) continue;
        // (a nested call on purpose: this function's own footprint stays small)
        tile_hits<false>(acc, wrow0 + 32u * (uint32_t)t, j0 + li, lane, ctx);
    }
}


// ---- panel-mark queue: what a false first-stage survivor costs -------------------------------------------------------------
// On real frame hashes 2.3e-4 of all pairs pass a 128-bit first stage (0.24 per 1024-pair tile) although only ~1e-8 are
// hits. Every MFMA-based way of settling them spends matrix-pipe time on a whole 32x32 tile for ONE pair: +1 MFMA per
// survivor in the register cascade (+11 % MFMAs and a dependent chain), 4 from memory in the fetch form. The queue form
// spends none: the panel loop is the fetch form's (v_alignbit shifts a tile's verdict into a per-lane mask), and once per
// PANEL the lanes whose mask is not empty push it, with their column, branch-free: a ballot, two counts, one predicated
// ds_write_b64 -- free next to 16 MFMAs. A mark stands for 8 accumulator registers of one tile (two marks per tile, one
// v_alignbit each; registers 4c .. 4c+3 are four CONSECUTIVE rows, ibrel + 8c + 0..3, see qrow_of); mark number idx = 2 t + q
// sits in bit (2 TILES - 1 - idx) of the mask. The queues are per wave (slots are handed out with a scalar counter, no
// atomics) but SETTLED by the whole workgroup right behind a super-panel barrier, on the VALU: the packed hashes differ
// exactly in the differing bits, so hamming = popcount(x ^ y); the filter looks at a mark's 8 rows against the column on the
// 128 bits the first stage did NOT see -- the work of ONE lane per entry instead of wave-wide instructions per surviving tile
// in the loop. A panel in which MANY lanes hold a survivor (a real cluster, the diagonal, two copies of one video) goes the
// tile route (panel_survivors: full recomputation + the tile-level hit handler with its video-key de-duplication).
// History of the forms that lost to this one (group-mask queue, 8-wave workgroups, 1 / 4 marks per tile): HISTORY.md.
constexpr uint32_t kQEntries = 1536;  // entries of all the workgroup's queues together (12 KB): 4 waves x 384
constexpr uint32_t kQMaxWaves = 8;
constexpr uint32_t kQDrainAt = 700;   // settle when the workgroup holds this many: one round of the filter (3 x 256) with the next super-panel's ~30 on top
constexpr uint32_t kQPanelLanes = 48; // a panel in which more lanes than this hold a survivor takes the tile route (a tile full of hits has 64)
constexpr int kQMarks = 2;            // marks per tile (1: 18.0, 2: 17.15, 4: 17.6 ms on frame hashes, profiles/r04_k2_queue_ablation.txt)
// entry: x = the lane's marks; y = column (absolute) << 1 | h (the lane's half: its rows start 4 h below the tile's
// first) -- hence n_pad < 2^31
__shared__ uint2 g_wave_queue[kQEntries];  // wave w's queue: [w * qcap, (w + 1) * qcap), qcap = kQEntries / waves
// Two sets of fill levels, used in turn: the words settle() of super-panel k reads behind its barrier are not the
// words publish() of super-panel k+1 writes, so a wave that is late to read can never see a sibling's NEWER level and come to
// a different drain decision (divergent barriers); the set of super-panel k is written again only by k+2, and the barrier of
// k+1 lies in between.
__shared__ __attribute__((aligned(16))) uint32_t g_wave_qn[2][kQMaxWaves];

__device__ __forceinline__ uint32_t sign_popc(const uint4& x, const uint4& y, uint32_t acc) {
    // FP4 images: equal magnitude bits cancel, only sign nibbles survive the XOR (rows beyond n are FP4 zeros -- filtered
    // by index); packed hashes: the plain Hamming distance
    return acc + __popc(x.x ^ y.x) + __popc(x.y ^ y.y) + __popc(x.z ^ y.z) + __popc(x.w ^ y.w);
}

// Which 128 bits the first stage sees (`sel`, the probe's choice, select[3]): 0 = bits 0..127, 1 = bits 128..255,
// 2 = bits 0..63 and 192..255 (round 5: on config 5's frame hashes this mix lets 3x fewer unrelated pairs through than
// either half, and a settled entry is what the pair-queue forms pay for). In 8-byte units of a packed hash (u0..u3) the
// first stage sees {u0,u1} / {u2,u3} / {u0,u3}; the settlement's filter looks at the OTHER 128 bits, which are one
// contiguous 16-byte block in every case: it starts at unit `ou` = 2 / 0 / 1. Only ou = 1 is not 16-byte aligned.
__device__ __forceinline__ uint32_t other_unit_of(uint32_t sel) { return sel == 0u ? 2u : sel == 1u ? 0u : 1u; }
template <class P>
__device__ __forceinline__ uint4 other_block(P db, size_t hash, uint32_t ou) {
    // ou is launch-uniform: the aligned cases keep their single 16-byte load (a scalar branch), the mix takes two 8-byte loads
    // from one line; P: a global or constant uint4*
    if ((ou & 1u) == 0u) return db[hash * 2u + (ou >> 1)];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&db[hash * 2u]) + 2u;
    const uint2 a = *reinterpret_cast<const uint2*>(w), b = *reinterpret_cast<const uint2*>(w + 2);
    return make_uint4(a.x, a.y, b.x, b.y);
}

// Row of accumulator register r of a lane whose rows start at ib (C/D layout of the 32x32 MFMA).
__device__ __forceinline__ uint32_t qrow_of(uint32_t ib, uint32_t r) { return ib + (r & 3u) + 8u * (r >> 2); }

// One candidate pair, all 256 bits, all checks, both sinks (rare: what the filter lets through).
__device__ __forceinline__ void settle_pair(const HitCtx& c, uint32_t i, uint32_t j) {
    const bool rect = c.rect != 0u;
    if (!(j < c.n && (rect ? i < c.nq : i < j))) return;
    uint32_t d = 0;
    if (c.db_t != nullptr) {  // launch-uniform
        const uint4 x0 = c.db_q[(size_t)i * 2u], x1 = c.db_q[(size_t)i * 2u + 1u];
        const uint4 y0 = c.db_t[(size_t)j * 2u], y1 = c.db_t[(size_t)j * 2u + 1u];
        d = sign_popc(x0, y0, sign_popc(x1, y1, 0u));
    } else {
        const uint4* __restrict__ pi = c.img_q + (size_t)i * 8u;
        const uint4* __restrict__ pj = c.img_t + (size_t)j * 8u;
        const uint32_t si = (i >> 1) & 7u, sj = (j >> 1) & 7u;
#pragma unroll 1
        for (uint32_t c0 = 0; c0 < 8u; c0 += 4u) {
#pragma unroll
            for (uint32_t ch = c0; ch < c0 + 4u; ++ch) d = sign_popc(pi[ch ^ si], pj[ch ^ sj], d);
        }
    }
    if (d > c.max_dist) return;
    if (c.group != nullptr && c.group[i] == (rect ? c.group_t[j] : c.group[j])) return;
    if (c.vs.set == nullptr) {
        append_pair_wg(c.out, c.cap, c.count, i, j, d);
    } else {
        hvd::sink_insert(c.vs, hvd::vkey_make(0u, i, (uint32_t)c.vs.vid_t[j]));
        hvd::sink_insert(c.vs, hvd::vkey_make(rect ? 1u : 0u, j, (uint32_t)c.vs.vid_q[i]));
    }
}

// The fill levels of the workgroup's queues as every thread sees them behind a barrier (wave-uniform scalars): pre[w] = number
// of entries in front of wave w's in the concatenation (wave 0's entries, wave 1's, ...), pre[kQMaxWaves] = all of them.
struct QCounts {
    uint32_t pre[kQMaxWaves + 1];
};
__device__ __forceinline__ QCounts load_qcounts(uint32_t waves, uint32_t par) {
    const uint4 lo = *reinterpret_cast<const uint4*>(&g_wave_qn[par][0]), hi = *reinterpret_cast<const uint4*>(&g_wave_qn[par][4]);
    const uint32_t raw[kQMaxWaves] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    QCounts c;
    uint32_t run = 0;
#pragma unroll
    for (uint32_t w = 0; w < kQMaxWaves; ++w) {
        c.pre[w] = run;
#if defined(__HIP_DEVICE_COMPILE__)
        run += w < waves ? (uint32_t)__builtin_amdgcn_readfirstlane((int)raw[w]) : 0u;
#else
        run += w < waves ? raw[w] : 0u;
#endif
    }
    c.pre[kQMaxWaves] = run;
    return c;
}
// entry number k (< pre[kQMaxWaves]) of the concatenation; *w_out = the wave it belongs to
__device__ __forceinline__ uint2* queue_entry(uint32_t k, const QCounts& c, uint32_t qcap, uint32_t* w_out) {
    uint32_t w = 0, base = 0;
#pragma unroll
    for (uint32_t i = 1; i < kQMaxWaves; ++i) {
        const bool past = k >= c.pre[i];  // (pre is non-decreasing; empty and absent waves share their successor's value)
        w = past ? i : w;
        base = past ? c.pre[i] : base;
    }
    *w_out = w;
    return &g_wave_queue[w * qcap + (k - base)];
}


// Settling the workgroup's queues takes two LEAF functions, both called by all of its threads behind a barrier (the
// caller puts another barrier behind them before anybody pushes again). Leaf, because a function that calls another keeps
// its own values in the high callee-saved registers, and every register a callee touches is one the kernel cannot hold a
// live value in across the call (a first version with nested calls: 167 VGPRs, 39 spills in the kernel's panel loop).
//
// 1. drain_filter_panel_wg filters a mark's 8 rows against the entry's column on the 128 bits the first stage did NOT see,
//    in the packed hashes: 16 B per hash; of unrelated pairs that agreed in one half, 2e-4 agree in the other. What this
//    costs is line REQUESTS and rounds, not bytes (profiles/r04_k2_queue_ablation.txt): a lane's 16 bytes from a line of its
//    own take the texture path a cycle each, and the other workgroups' panel prefetch queues behind them. So (a) the
//    workgroup's own rows -- 1024 x 16 B -- are first copied, coalesced, into the panel buffer that is free at this point
//    (rows_lds: 16 KB), and only an entry's COLUMN is a gather; (b) a thread takes three entries per round with their loads
//    in flight together, and the queues are settled when they hold about one round's worth (kQDrainAt). The filter writes
//    back the marks that hold a HIT (x = 0: nothing left). Returns non-zero if this thread left work for
// 2. settle_marked_panel_wg, which walks the same entries again and reports every row that passes in full (settle_pair).
//    Without packed hashes (image-only callers) every row of every mark is its work.
// geom: waves | tiles << 4 | entries per wave << 8 | set of fill levels << 24
#if defined(__HIP_DEVICE_COMPILE__)
// Does any of four consecutive rows pass? Every lane's four rows start at a multiple of 64 bytes, so a ds_read_b128 that
// takes the r-th row of every lane finds all of them in 4 of the 16 groups of four banks: the lanes walk their rows in an
// order rotated by the lane number instead (which row passed is not asked here).
__device__ __forceinline__ uint32_t filter_rows4(const __attribute__((address_space(3))) uint4* rows, uint32_t rot,
                                                 const uint4& col, uint32_t max_dist) {
    uint32_t pass = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4u; ++r) pass |= sign_popc(rows[(r + rot) & 3u], col, 0u) <= max_dist ? 1u << r : 0u;
    return pass;
}
#endif
__device__ __noinline__ uint32_t drain_filter_panel_wg(const HitCtx* __restrict__ ctx, uint32_t geom_v, uint32_t row0_v,
                                                       uint32_t other_half_v, uint32_t tid, uint4* rows_generic) {
#if defined(__HIP_DEVICE_COMPILE__)
    auto* rows_lds = (__attribute__((address_space(3))) uint4*)rows_generic;  // (ds_* instead of flat_* accesses)
    const uint32_t geom = (uint32_t)__builtin_amdgcn_readfirstlane((int)geom_v);
    const uint32_t row0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)row0_v);
    const uint32_t ohw = (uint32_t)__builtin_amdgcn_readfirstlane((int)other_half_v);
    const auto* cc = (const __attribute__((address_space(4))) HitCtx*)uniform_u64((unsigned long long)ctx);
    const uint32_t waves = geom & 15u, tiles = (geom >> 4) & 15u, qcap = (geom >> 8) & 0xFFFu, nthreads = 64u * waves;
    constexpr uint32_t qg = (uint32_t)kQMarks, qshift = qg >> 1, chunks = 4u >> qshift;  // 4-row chunks per mark
    const uint32_t top = tiles * qg - 1u;
    const QCounts qc = load_qcounts(waves, (geom >> 24) & 1u);
    const uint32_t total = qc.pre[kQMaxWaves];
    const uint4* __restrict__ db_q = cc->db_q;
    const uint4* __restrict__ db_t = cc->db_t;
    const uint32_t max_dist = cc->max_dist;
    if (db_t == nullptr) return 1u;  // launch-uniform: no packed hashes, every mark stays to do
    const uint32_t ou = ohw & 3u, wrows = ohw >> 8;  // first unit of the other 128 bits | rows per wave << 8
    const uint32_t last = (cc->rect != 0u ? cc->nq : cc->n) - 1u;  // rows beyond it are padding: never queued, never read
    constexpr int E = 3;  // entries a thread takes per round (2 .. 5 with matching kQDrainAt: within 1.5 %)
    uint32_t left = 0;
    bool staged = false;
#pragma unroll 1
    for (uint32_t k0 = 0; k0 < total; k0 += nthreads * E) {
        uint32_t ex[E], lanerel[E];
        uint4 col[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const uint32_t k = k0 + tid + nthreads * (uint32_t)u;
            ex[u] = 0u;
            lanerel[u] = 0u;
            col[u] = make_uint4(0u, 0u, 0u, 0u);
            if (k < total) {
                uint32_t w;
                const uint2 e = *queue_entry(k, qc, qcap, &w);
                ex[u] = e.x;
                lanerel[u] = 4u * (e.y & 1u) + wrows * w;  // the lane's first row, relative to row0
                col[u] = other_block(db_t, (size_t)(e.y >> 1), ou);
            }
        }
        // the workgroup's own rows (coalesced) are requested BEHIND the first round's column gathers, so that the two memory
        // round trips of a settlement run side by side
        if (!staged) {
#pragma unroll
            for (uint32_t r = tid; r < kSuper * 8u; r += nthreads) rows_lds[r] = other_block(db_q, (size_t)min(row0 + r, last), ou);
            __syncthreads();
            staged = true;
        }
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const uint32_t k = k0 + tid + nthreads * (uint32_t)u;
            if (k >= total) continue;
            uint32_t m = ex[u], kept = 0;
#pragma unroll 1
            while (m != 0u) {
                const uint32_t bit = 31u - (uint32_t)__clz((int)m);
                m &= ~(1u << bit);
                const uint32_t idx = top - bit, t = idx >> qshift, q = idx & (qg - 1u);
                const uint32_t ib = lanerel[u] + 32u * t + 8u * chunks * q;
                uint32_t pass = 0;
#pragma unroll 1
                for (uint32_t c = 0; c < chunks; ++c) {
                    uint32_t p4 = filter_rows4(rows_lds + ib + 8u * c, tid, col[u], max_dist);
                    // A row that passes here by chance (2.3e-4 of them: one or two per settlement) would send the whole
                    // workgroup through settle_marked_panel_wg -- another walk over the queues and another memory round trip
                    // with everybody waiting. The lane that found it looks at the first stage's half as well, on the spot;
                    // what it keeps is a hit (A/B: -3.4 %).
                    if (__builtin_expect(p4 != 0u, 0)) {
                        uint32_t w2;
                        const uint32_t j = queue_entry(k, qc, qcap, &w2)->y >> 1;
                        // (all 256 bits from the packed hashes: the first stage's 128 are not one block for every selection)
                        const uint4 c0 = db_t[(size_t)j * 2u], c1 = db_t[(size_t)j * 2u + 1u];
                        uint32_t hit = 0;
#pragma unroll 1
                        while (p4 != 0u) {
                            const uint32_t r = (uint32_t)__ffs((int)p4) - 1u;
                            p4 &= p4 - 1u;
                            const uint32_t rowrel = ib + 8u * c + ((r + tid) & 3u);
                            const size_t ri = (size_t)min(row0 + rowrel, last);
                            if (sign_popc(db_q[ri * 2u], c0, sign_popc(db_q[ri * 2u + 1u], c1, 0u)) <= max_dist) hit = 1u;
                        }
                        p4 = hit;
                    }
                    pass |= p4;
                }
                if (pass != 0u) kept |= 1u << bit;
            }
            left |= kept;
            uint32_t w_;
            queue_entry(k, qc, qcap, &w_)->x = kept;
        }
    }
    return left;
#else
    (void)ctx; (void)geom_v; (void)row0_v; (void)other_half_v; (void)tid; (void)rows_generic;
    return 0u;
#endif
}

__device__ __noinline__ void settle_marked_panel_wg(const HitCtx* __restrict__ ctx, uint32_t geom_v, uint32_t row0_v,
                                                    uint32_t other_half_v, uint32_t tid, const uint4* rows_generic) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto* rows_lds = (const __attribute__((address_space(3))) uint4*)rows_generic;
    const uint32_t geom = (uint32_t)__builtin_amdgcn_readfirstlane((int)geom_v);
    const uint32_t row0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)row0_v);
    const uint32_t ohw = (uint32_t)__builtin_amdgcn_readfirstlane((int)other_half_v);
    const uint32_t ou = ohw & 3u, wrows = ohw >> 8;
    const uint32_t waves = geom & 15u, tiles = (geom >> 4) & 15u, qcap = (geom >> 8) & 0xFFFu, nthreads = 64u * waves;
    constexpr uint32_t qg = (uint32_t)kQMarks, qshift = qg >> 1, nrows = 16u >> qshift;
    const uint32_t top = tiles * qg - 1u;
    const QCounts qc = load_qcounts(waves, (geom >> 24) & 1u);
    const HitCtx c = load_ctx(ctx);
    const uint32_t total = qc.pre[kQMaxWaves];
    const bool packed = c.db_t != nullptr;
#pragma unroll 1
    for (uint32_t k = tid; k < total; k += nthreads) {
        uint32_t w;
        const uint2 e = *queue_entry(k, qc, qcap, &w);
        uint32_t m = e.x;
        if (m == 0u) continue;
        const uint32_t lanerel = 4u * (e.y & 1u) + wrows * w, j = e.y >> 1;
        const uint4 col = packed ? other_block(c.db_t, (size_t)j, ou) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 1
        while (m != 0u) {
            const uint32_t bit = 31u - (uint32_t)__clz((int)m);
            m &= ~(1u << bit);
            const uint32_t idx = top - bit, t = idx >> qshift, q = idx & (qg - 1u);
            const uint32_t ibrel = lanerel + 32u * t, r0 = nrows * q;
#pragma unroll 1
            for (uint32_t r = r0; r < r0 + nrows; ++r) {
                const uint32_t rr = qrow_of(ibrel, r);
                if (packed && sign_popc(rows_lds[rr], col, 0u) > c.max_dist) continue;
                settle_pair(c, row0 + rr, j);
            }
        }
    }
#else
    (void)ctx; (void)geom_v; (void)row0_v; (void)other_half_v; (void)tid; (void)rows_generic;
#endif
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

// The all-pairs kernel. Three numbers select the form (all exact, all bit-identical in their output):
//   TILES  query tiles of 32 hashes per wave (4 waves per workgroup);
//   S1     k-steps of the first stage: 4 = the whole 256-bit dot product at once; 2 = 128 bits, which bound the distance
//          from below (a partial distance above the tolerance implies a full one above it), and only the rare tile that
//          survives goes on to the other 128 bits;
//   NBR    k-steps of the query fragments that live in registers: with S1 = 2 either 2 (the survivor's other half is
//          fetched from memory: cheapest first stage, best when survivors are very rare -- uniform random hashes) or 4 (the
//          second stage runs out of registers, 128 -> 192 -> 256 bits: robust when most tiles hold several survivors);
//   QUEUE  (NBR = S1 = 2 only) first-stage survivors are settled pair by pair on the VALU (panel-mark queue, above) instead
//          of tile by tile on the matrix pipe: the form for data on which false survivors are common (real frame hashes).
// The forms that have a job (round 6; the others are in HISTORY.md): 8 = <8,4,4> the 256-bit reference; 9 = <8,2,2> fetch
// (uniform data: the headline); 12 = <4,4,2> register cascade (dense data); 18 = <8,2,2,QUEUE> (frame hashes).
// RECT = false: img_q == img (one set, strict upper triangle). RECT = true: rows come from the query image img_q (nq
// hashes), candidates from the target image img (n hashes), full rectangle.
// The query fragments are NEGATED and the accumulators start at (threshold - 1), so that a hit is a set sign bit (or16_bits).
template <int TILES, int NBR, int S1, bool RECT, bool QUEUE = false>
// (the 4-tile register form is held to 3 waves per SIMD = 168 VGPRs: with the pre-read fragment it would take 170)
__global__ __launch_bounds__(256, ((TILES == 4 && NBR == 4 && S1 == 2) || QUEUE) ? 3 : 2) void k_allpairs_mfma(const uint4* __restrict__ img, uint32_t n, uint32_t n_pad,
                                                          uint32_t max_dist, uint32_t col_chunk, uint32_t rank,
                                                          uint32_t world, const uint4* __restrict__ img_q,
                                                          const HitCtx* __restrict__ ctx,
                                                          const uint32_t* __restrict__ select, uint32_t select_id, uint32_t sel_arg) {
    static_assert(S1 == 2 || S1 == 4, "first stage = 128 or 256 bits");
    static_assert(NBR >= S1 && (NBR == 2 || NBR == 4), "register-resident k-steps");
    static_assert(!QUEUE || (NBR == 2 && S1 == 2 && TILES <= 8), "the pair queue belongs to the 128-bit fetch form");
    constexpr int WAVES = 4;
    constexpr uint32_t WROWS = 32u * TILES, ROWS = (uint32_t)WAVES * WROWS, NT = 64u * WAVES;
    constexpr int QG = kQMarks;
    static_assert(TILES * QG <= 32, "a lane's marks are one word");
    // QUEUE: entries per wave, and the most one wave can add between two barriers
    constexpr uint32_t QCAP = kQEntries / WAVES, QSUPERMAX = (kSuper / 32) * kQPanelLanes;
    static_assert(!QUEUE || QCAP >= QSUPERMAX + 64, "a wave's queue must take a super-panel's worth on top of a carry-over");
    __shared__ uint4 lds0[kSuper * 8], lds1[kSuper * 8];

    // data-dependent choice between the forms of this kernel (launch_auto): all are launched, the probe's verdict lets one
    // of them run
    if (select != nullptr && *select != select_id) return;
    // Clock telemetry (round 6): one workgroup in eight brackets its lifetime with the shader-cycle counter (s_memtime) and the
    // constant-rate counter (s_memrealtime) and adds both deltas to the context's accumulators -- effective shader clock of
    // a pass = cycles / ticks x the tick rate, averaged over the sampled workgroups' lifetimes (hvd_debug_get "mfma_pass_khz").
    // The start values wait in LDS (16 B), not in SGPRs: this kernel has no scalar register to spare (106 of 106, the queue
    // form already parks 70 in VGPR lanes), and four more held across the tile cost form 9 its third resident wave.
    if (threadIdx.x == 0u) {
        const bool sample = ((blockIdx.x + blockIdx.y) & 7u) == 0u;
        g_clk_t0[0] = sample ? (unsigned long long)__builtin_readcyclecounter() | 1ull : 0ull;  // (0 = not sampled)
        g_clk_t0[1] = sample ? (unsigned long long)__builtin_amdgcn_s_memrealtime() : 0ull;
    }

    // Which 128 bits the first stage sees is the probe's choice too (select[3]): the image keeps bits 0..127 in chunks
    // 0..3 and bits 128..255 in chunks 4..7, so "the other half first" is chunk ^ 4 in every fragment address -- a
    // launch-uniform XOR into the slot swizzle. The full-distance paths sum over all eight chunks and do not care.
    // Round 5: a third selection, bits 0..63 + 192..255 (other_unit_of, above): k-step s reads chunks 2s, 2s+1 XOR selx_s with
    // selx_0 = selx_2, selx_1 = selx_3 in {0, 4}, so that the four steps still cover every chunk once.
    const uint32_t sel = select != nullptr ? (select[3] & 3u) : sel_arg;
    const uint32_t selx0 = sel == 1u ? 4u : 0u, selx1 = sel != 0u ? 4u : 0u;

    // Tile (rb, cb) belongs to rank (rb + cb) mod world. Round 5: only a rank's OWN tiles are launched (grid.y = ceil(n_cb /
    // world)): workgroup y of row block rb takes column chunk y * world + (rank - rb) mod world. Launching every tile and
    // letting seven of eight workgroups return cost ~0.2 ns each -- 0.2 ms per form and pass at 2.8 M hashes, three forms per
    // auto pass: 4 % of an 18 ms step at N = 8, straight out of the scaling efficiency.
    const uint32_t rb = blockIdx.x;
    const uint32_t cb = world > 1u ? blockIdx.y * world + (rank + world - rb % world) % world : blockIdx.y;
    const uint32_t row0 = rb * ROWS;
    const uint32_t col0 = cb * col_chunk;
    if (col0 >= n_pad) return;  // (the last residues of a row block may lie beyond the last chunk)
    const uint32_t col1 = min(col0 + col_chunk, n_pad);
    if (!RECT && min(col1, n) <= row0 + 1u) return;  // tile entirely on/below the diagonal
    const uint4* __restrict__ imgq = RECT ? img_q : img;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, li = lane & 31u, h = lane >> 5;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wrow0 = row0 + wave * WROWS;
    if (tid == 0u) g_wg_npairs = 0u;  // (ordered before any hit by the barrier behind the first staged super-panel)

    // A fragments: query hash (wrow0 + 32t + li), chunk 2s+h, for the whole tile.
    v4i a[TILES][NBR];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const uint32_t hash = wrow0 + 32u * t + li;  // < n_pad by construction
#pragma unroll
        for (int s = 0; s < NBR; ++s) {
            const uint4 q = imgq[(size_t)hash * 8u + (img_slot(hash, 2u * s + h) ^ ((s & 1) ? selx1 : selx0))];
            // negated: flip the sign bit of every e2m1 nibble (see or16_bits)
            const uint32_t fl = 0x88888888u;
            a[t][s] = v4i{(int)(q.x ^ fl), (int)(q.y ^ fl), (int)(q.z ^ fl), (int)(q.w ^ fl)};
        }
    }

    // first stage: acc = c1 - dot over 64*S1 bits, hit candidate <=> acc < 0; the second stage of the register form
    // goes on to acc = c1 - dot256, a hit <=> dot256 >= 256 - 2*max_dist <=> acc <= c1 - thr_full = -129 (S1 = 2)
    const float c1 = (float)(64 * S1) - 2.0f * (float)max_dist - 1.0f;
    const float hit2 = -128.5f;
    const float hit192 = -64.5f;  // after 192 bits: dot192 >= 192 - 2*max_dist <=> acc <= c1 - that = -65
    // marks' = marks << 1 | verdict(acc): v_alignbit_b32 takes the sign bit of the OR straight into the mask
    auto stage1_mark = [&](uint32_t marks, const v16f& acc) -> uint32_t {
        return __builtin_amdgcn_alignbit(marks, (uint32_t)or16_bits(acc), 31);
    };
    auto stage1_hit = [&](const v16f& acc) { return __any(or16_bits(acc) < 0); };
    auto stage2_hit = [&](const v16f& acc) { return __any(min16(acc) < hit2); };
    auto stage192_hit = [&](const v16f& acc) { return __any(min16(acc) < hit192); };

    // candidates <= row0 cannot pair with rows >= row0 (i<j): start at the super-panel holding row0+1
    const uint32_t j0 = RECT ? col0 : max(col0, (row0 + 1u) & ~(uint32_t)(kSuper - 1));
    const uint32_t nsp = (col1 - j0) / kSuper;  // col0, col1, j0 are multiples of kSuper

    const v16f zero = {c1, c1, c1, c1, c1, c1, c1, c1, c1, c1, c1, c1, c1, c1, c1, c1};  // the accumulators' start value
    // QUEUE: index of this wave's next free slot in the (flattened) queue array (wave-uniform) -- the fill level and the
    // wave's base in one scalar -- and this lane's share of an entry's y word: (column << 1) | h
    uint32_t qidx = wave * QCAP;
    const uint32_t qcol = (li << 1) | h;

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
            const uint32_t swb = (cl >> 1) & 7u;  // jsp is a multiple of 128: same swizzle as the global index
            const uint32_t sw0 = swb ^ selx0, sw1 = swb ^ selx1;  // k-steps 0, 2 / 1, 3 (the selection, above)
            v4i b[S1];
#pragma unroll
            for (int s = 0; s < S1; ++s) b[s] = as_v4i(base[(2u * s + h) ^ ((s & 1) ? sw1 : sw0)]);
            // register cascade form: the 192-bit step's B fragment is read WITH the panel's first two, so that a first-stage
            // survivor goes straight to its MFMA instead of waiting out an LDS round trip first (on frame hashes most
            // panels have one; same-box A/B on structured hashes: -2.5 %)
            v4i b192 = b[0];
            if constexpr (NBR == 4 && S1 == 2) b192 = as_v4i(base[(4u + h) ^ sw0]);

            // two accumulator sets: the MFMAs of tile t+1 are issued before the max tree of tile t
            if constexpr (QUEUE) {
                // the fetch form's loop; what it marks is pushed once per panel, with no branch (panel-mark queue, above)
                uint32_t marks = 0;  // bit (TILES * QG - 1 - (t * QG + q)) <-> registers 16 / QG * q ... of tile t
                auto part_mark = [&](uint32_t mk, const v16f& acc) -> uint32_t {
#pragma unroll
                    for (int q = 0; q < QG; ++q) {
                        int x = 0;
#pragma unroll
                        for (int r = 0; r < 16 / QG; ++r) x |= __float_as_int(acc[16 / QG * q + r]);
                        mk = __builtin_amdgcn_alignbit(mk, (uint32_t)x, 31);
                    }
                    return mk;
                };
                v16f cur = tile_dot<0, S1>(a[0], b, zero);
#pragma unroll
                for (int t = 1; t < TILES; ++t) {
                    const v16f nxt = tile_dot<0, S1>(a[t], b, zero);
                    marks = part_mark(marks, cur);
                    cur = nxt;
                }
                marks = part_mark(marks, cur);
                const unsigned long long act = __ballot(marks != 0u);
                const uint32_t nl = (uint32_t)__builtin_popcount((uint32_t)act) + (uint32_t)__builtin_popcount((uint32_t)(act >> 32));
                const bool dense = nl > kQPanelLanes;
                // (a second entry for the lanes that hold two marks -- 6 % of those that push; a settling wave is as slow as its
                // lane with the most -- was measured: +1.5 %, the extra ballot and push in this loop cost more than they save)
                // (the array stays NAMED in the store: through a bare LDS address the compiler cannot tell it from the panel
                // buffers and waits for the prefetch in flight, vmcnt(0), first)
                if (marks != 0u && !dense) {
                    const uint32_t mb = __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
                    g_wave_queue[qidx + mb] = make_uint2(marks, ((jsp + 32u * p) << 1) + qcol);
                }
                qidx += dense ? 0u : nl;
                if (__builtin_expect(dense, 0)) {
                    uint32_t tm = 0;  // one bit per tile for the tile route
#pragma unroll
                    for (int t = 0; t < TILES; ++t) tm |= ((marks >> (QG * (TILES - 1 - t))) & ((1u << QG) - 1u)) != 0u ? 1u << (TILES - 1 - t) : 0u;
                    panel_survivors<TILES>(tm, imgq, panel, wrow0, jsp + 32u * p, lane, ctx);
                }
            } else if constexpr (NBR == 2) {
                // survivors are only noted -- one VALU op per tile shifts the tile's verdict (the sign of the OR, or the
                // compare's result) into a per-lane mask -- and dealt with after the panel, when no accumulator is live
                // any more: the handler's registers add to whatever is live across its call. One compare per PANEL.
                uint32_t marks = 0;  // bit (TILES-1-t) <-> tile t
                v16f cur = tile_dot<0, S1>(a[0], b, zero);
#pragma unroll
                for (int t = 1; t < TILES; ++t) {
                    const v16f nxt = tile_dot<0, S1>(a[t], b, zero);
                    marks = stage1_mark(marks, cur);
                    cur = nxt;
                }
                marks = stage1_mark(marks, cur);
                if (__builtin_expect(__any(marks != 0u), 0)) panel_survivors<TILES>(marks, imgq, panel, wrow0, jsp + 32u * p, lane, ctx);
            } else {
                // each tile is judged on its own: a survivor's second stage runs out of registers at once, and only a
                // tile with a real hit (all 256 bits) calls the handler
                auto survivor = [&](const int t, v16f acc) {
                    if (S1 == 2) {
                        // cascade 128 -> 192 -> 256 bits (every partial distance bounds the full one from below): on dense
                        // data most (wave, panel) steps see a false survivor -- it costs ONE more MFMA, and only what also
                        // survives 192 bits a second
                        acc = mfma_fp4(a[t][2], b192, acc);
                        if (!stage192_hit(acc)) return;
                        acc = mfma_fp4(a[t][3], as_v4i(base[(6u + h) ^ sw1]), acc);
                        if (!stage2_hit(acc)) return;
                    }
                    tile_hits<true>(acc, wrow0 + 32u * (uint32_t)t, jsp + cl, lane, ctx);
                };
                v16f cur = tile_dot<0, S1>(a[0], b, zero);
#pragma unroll
                for (int t = 1; t < TILES; ++t) {
                    const v16f nxt = tile_dot<0, S1>(a[t], b, zero);
                    if (__builtin_expect(stage1_hit(cur), 0)) survivor(t - 1, cur);
                    cur = nxt;
                }
                if (__builtin_expect(stage1_hit(cur), 0)) survivor(TILES - 1, cur);
            }
        }
    };

    // QUEUE: every wave publishes its fill level in front of a super-panel barrier; behind it the workgroup decides -- on
    // the same four numbers -- whether to settle the queues now.
    auto publish = [&](const uint32_t par) {
        if constexpr (QUEUE) {
            if (lane == 0u) g_wave_qn[par][wave] = qidx - wave * QCAP;
        }
    };
    auto settle = [&](const bool final, uint4* free_panel, const uint32_t par) {
        if constexpr (QUEUE) {
            const QCounts qc = load_qcounts(WAVES, par);
            const uint32_t sum = qc.pre[kQMaxWaves];
            uint32_t mx = 0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) mx = max(mx, qc.pre[w + 1] - qc.pre[w]);
            if (final ? sum != 0u : (sum >= kQDrainAt || mx > QCAP - QSUPERMAX)) {
                const uint32_t GEOM = (uint32_t)WAVES | ((uint32_t)TILES << 4) | (QCAP << 8) | (par << 24);
                const uint32_t left = drain_filter_panel_wg(ctx, GEOM, row0, other_unit_of(sel) | (WROWS << 8), wave * 64u + lane, free_panel);
                if (__builtin_expect(__any(left != 0u), 0)) settle_marked_panel_wg(ctx, GEOM, row0, other_unit_of(sel) | (WROWS << 8), wave * 64u + lane, free_panel);
                qidx = wave * QCAP;
                __syncthreads();  // nobody pushes (or publishes) again before everybody has read the queues
            }
        }
    };

    stage_super_panel(img + (size_t)j0 * 8u, lds0, wave, lane);
    __syncthreads();

    for (uint32_t sp = 0; sp < nsp; sp += 2) {
        const uint32_t jsp = j0 + sp * kSuper;
        // lds1 was last read in iteration sp-1, which every wave left through a barrier
        if (sp + 1u < nsp) stage_super_panel(img + (size_t)(jsp + kSuper) * 8u, lds1, wave, lane);
        process(lds0, jsp);
        publish(0u);
        __syncthreads();  // (drains the in-flight global->LDS loads with vmcnt(0) first)
        if (sp + 1u >= nsp) break;
        settle(false, lds0, 0u);  // (lds0 has just been used up and is not refilled before the settlement is over)
        if (sp + 2u < nsp) stage_super_panel(img + (size_t)(jsp + 2u * kSuper) * 8u, lds0, wave, lane);
        process(lds1, jsp + kSuper);
        publish(1u);
        __syncthreads();
        if (sp + 2u < nsp) settle(false, lds1, 1u);
    }
    // the last super-panel's levels: set 0 if their number is odd (nsp >= 1: the tile holds columns beyond its first row)
    settle(true, lds0, (nsp & 1u) ^ 1u);  // (nothing is in flight any more: both buffers are free)
    // every path leaves the loop through a barrier: all hits of this workgroup are in LDS now
    flush_pairs_wg(ctx, wave * 64u + lane, NT);
}

// Probe for the data-dependent choice of the kernel form: over a strided sample of the two images (up to 4096 rows
// x 4096 columns), how often do the first 128 bits of two hashes agree to within the tolerance? In the FP4 image a
// differing bit is a differing sign nibble, so the partial distance is popcount(x ^ y) over chunks 0..3.
// One lane per sample row, 64 sample columns per workgroup broadcast from LDS. select[1] += survivors.
constexpr uint32_t kProbeRows = 4096, kProbeCols = 4096;

// (round 4: 64 sample columns per workgroup instead of 256 -- 1024 workgroups instead of 256, four waves per SIMD instead of a
// lone one whose dependent popcount chain issues every ~5 cycles: 78 -> ~25 us per pass; sampled hashes are real ones, whose
// images differ in sign nibbles only, so the XOR needs no mask)
constexpr uint32_t kProbeColsPerWg = 64;
struct ProbeRule {  // survivors among `pairs` sampled pairs -> form (probe_decide, below)
    uint64_t pairs;
    uint32_t pairs_per_step, id_rare, id_mid, id_often;
    float mid_max_per_tile;
    int force_sel;  // >= 0: the first-stage selection is not the probe's to choose (hvd_debug_set "mfma_force_sel": tests, A/B runs)
};
__device__ void probe_decide(uint32_t* __restrict__ select, uint32_t lo, uint32_t hi, uint32_t mix, const ProbeRule& rule);

__global__ __launch_bounds__(256) void k_prefilter_probe(const uint4* __restrict__ img_q, uint32_t nq,
                                                         const uint4* __restrict__ img_t, uint32_t nt, uint32_t max_dist,
                                                         uint32_t* __restrict__ select, const ProbeRule rule) {
    __shared__ uint4 cols[kProbeColsPerWg][8];
    const uint32_t rows = min(nq, kProbeRows), ncols = min(nt, kProbeCols);
    const uint32_t rstride = nq / rows, cstride = nt / ncols;
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    const uint32_t ri = min(r, rows - 1u) * rstride;
    uint4 q[8];
#pragma unroll
    for (uint32_t c = 0; c < 8; ++c) q[c] = img_q[(size_t)ri * 8u + img_slot(ri, c)];
    // columns sit half a stride off the rows so that, in the symmetric form, a sample row never meets itself
    {
        const uint32_t cl = threadIdx.x >> 2, k0 = (threadIdx.x & 3u) * 2u;  // four threads fetch one column's eight chunks
        const uint32_t c = blockIdx.y * kProbeColsPerWg + cl;
        const uint32_t ci = min(min(c, ncols - 1u) * cstride + cstride / 2u, nt - 1u);
        cols[cl][k0] = img_t[(size_t)ci * 8u + img_slot(ci, k0)];
        cols[cl][k0 + 1u] = img_t[(size_t)ci * 8u + img_slot(ci, k0 + 1u)];
    }
    __syncthreads();
    const uint32_t c0 = blockIdx.y * kProbeColsPerWg;
    const uint32_t m = c0 >= ncols ? 0u : min(kProbeColsPerWg, ncols - c0);
    // survivors of a first stage over bits 0..127 / over bits 128..255 / over bits 0..63 + 192..255 (per 64-bit unit u0..u3:
    // the same 32 popcounts as before, three sums instead of two)
    uint32_t cnt = 0, cnt_hi = 0, cnt_mix = 0;
    for (uint32_t k = 0; k < m; ++k) {
        const uint32_t u0 = sign_popc(q[0], cols[k][0], sign_popc(q[1], cols[k][1], 0u)), u1 = sign_popc(q[2], cols[k][2], sign_popc(q[3], cols[k][3], 0u));
        const uint32_t u2 = sign_popc(q[4], cols[k][4], sign_popc(q[5], cols[k][5], 0u)), u3 = sign_popc(q[6], cols[k][6], sign_popc(q[7], cols[k][7], 0u));
        cnt += u0 + u1 <= max_dist ? 1u : 0u;
        cnt_hi += u2 + u3 <= max_dist ? 1u : 0u;
        cnt_mix += u0 + u3 <= max_dist ? 1u : 0u;
    }
    if (r >= rows) cnt = cnt_hi = cnt_mix = 0;
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off);
        cnt_hi += __shfl_down(cnt_hi, off);
        cnt_mix += __shfl_down(cnt_mix, off);
    }
    // one pair of atomics per WORKGROUP (same-address device atomics take ~8 ns each: per wave they were 65 us of a probe over
    // frame hashes), and the workgroup that finishes last turns the two sums into the decision -- one launch less per pass (a
    // one-lane kernel costs ~5 us plus the gap in front of it). No fence anywhere: the sums travel in atomics, which are
    // performed at the memory side, and a workgroup takes its ticket (select[4], zeroed by k_set_hit_ctx with the rest) only
    // after its own additions have RETURNED.
    __shared__ uint32_t part[3][4];
    if ((threadIdx.x & 63u) == 0u) {
        part[0][threadIdx.x >> 6] = cnt;
        part[1][threadIdx.x >> 6] = cnt_hi;
        part[2][threadIdx.x >> 6] = cnt_mix;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t c_lo = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        const uint32_t c_hi = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        const uint32_t c_mix = part[2][0] + part[2][1] + part[2][2] + part[2][3];
        uint32_t seen = 0;
        if (c_lo) seen += atomicAdd(&select[1], c_lo);
        if (c_hi) seen += atomicAdd(&select[2], c_hi);
        if (c_mix) seen += atomicAdd(&select[5], c_mix);
        asm volatile("" ::"v"(seen));  // (the returning form, and its result waited for)
        if (atomicAdd(&select[4], 1u) == gridDim.x * gridDim.y - 1u)
            probe_decide(select, atomicAdd(&select[1], 0u), atomicAdd(&select[2], 0u), atomicAdd(&select[5], 0u), rule);
    }
}

// The hit handler's launch-uniform arguments live in device memory (written by this one-lane kernel in stream order
// in front of the pass), so that the rare handler call passes one pointer instead of ~30 argument registers that the
// fast path's register allocation would have to keep clear.
// (select != nullptr: the auto variant's launch -- the probe's words are cleared in the same launch)
__global__ void k_set_hit_ctx(HitCtx* __restrict__ dst, const HitCtx src, uint32_t* __restrict__ select) {
    *dst = src;
    if (select != nullptr) {
#pragma unroll
        for (int k = 0; k < 6; ++k) select[k] = 0u;
    }
}

// the clock telemetry's accumulators (HitCtx::clk): cycles, ticks, sampled workgroups, spare
__global__ void k_clk_reset(unsigned long long* clk) {
    clk[0] = clk[1] = clk[2] = clk[3] = 0ull;
}

// survivors among `pairs` sampled pairs -> form. The fetch form (id_rare) pays ~6 panel steps per surviving TILE: right when
// survivors are real near-duplicates (uniform random hashes). The pair-queue form (id_mid) pays the settlement of one queue
// entry per surviving lane and nothing on the matrix pipe: right for real frame hashes, whose first 128 bits agree within the
// tolerance for ~2e-4 of unrelated pairs. The register form (id_often) pays one MFMA per surviving tile however many pairs
// survive in it: right when most tiles hold several survivors (a library whose hashes barely differ in either half).
__device__ void probe_decide(uint32_t* __restrict__ select, uint32_t lo, uint32_t hi, uint32_t mix, const ProbeRule& rule) {
    // first the selection: the 128 bits that let the fewest unrelated pairs through (ties and near-ties stay with bits 0..127,
    // so that uniform data always runs the same configuration; an alternative has to be 20 % better than what it replaces);
    // then the form, from that selection's rate
    uint32_t sel = 0u, best = lo;
    if ((double)hi * 1.25 < (double)best) {
        sel = 1u;
        best = hi;
    }
    if ((double)mix * 1.25 < (double)best) {
        sel = 2u;
        best = mix;
    }
    if (rule.force_sel >= 0) {
        sel = (uint32_t)rule.force_sel;
        best = sel == 0u ? lo : sel == 1u ? hi : mix;
    }
    select[3] = sel;
    const double rate = rule.pairs ? (double)best / (double)rule.pairs : 0.0;
    uint32_t form = rule.id_rare;
    if (rate * (double)rule.pairs_per_step > 0.01)
        form = (rule.id_mid != 0u && rate * 1024.0 <= (double)rule.mid_max_per_tile) ? rule.id_mid : rule.id_often;
    select[0] = form;
}

}  // namespace

namespace hvd {

static uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

constexpr int kClkWord = 192;  // 32-bit word offset of the clock telemetry's accumulators in a context's select buffer (byte 768)


// auto variant (13): the form for data with common false survivors (18 = panel-mark queue; 0 = none, i.e. round 3's two-way
// choice) and the survivor density (per 1024-pair tile, as the probe estimates it) up to which it is preferred over the register form
uint32_t g_mfma_queue_packed = 1;  // 0: the pair queue settles from the FP4 images even when packed hashes are at hand (tests)
uint32_t g_mfma_auto_mid = 18;
int g_mfma_force_sel = -1;  // hvd_debug_set "mfma_force_sel": -1 the probe chooses (explicit forms: bits 0..127); 0 | 1 | 2 forced
uint32_t g_mfma_auto_mid_max_x100 = 500;  // (scripts/gpu_k2_rate_sweep.py: the panel-mark queue takes 0.61-0.67 of the register form's time
                                          // at 1-2.7 survivors per tile, 0.79 at 4, 0.88 at 5.3, 1.23 at 8; round 4's first queue form, 15,
                                          // won up to ~1 and the boundary was 1.3)


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

struct MfmaForm {
    int tiles, nbr, s1, waves = 4;
};
// variants: 8 = 256 bits at once (the reference form); 9 = 128-bit first stage, survivors fetch their other half (uniform
// data); 12 = 128-bit first stage, second stage out of registers (4 tiles; dense data); 18 = 128-bit first stage, survivors
// through the panel-mark queue (frame hashes); 13 = 9, 18 or 12, chosen per launch by the probe.
static bool mfma_form(int variant, MfmaForm* f) {
    switch (variant) {
        case 8: *f = {8, 4, 4}; return true;
        case 9: case 13: case 18: *f = {8, 2, 2}; return true;
        case 12: *f = {4, 4, 2}; return true;
        default: return false;
    }
}

bool allpairs_mfma_geometry(uint32_t n, int variant, uint32_t* rows_per_block, uint32_t* col_chunk) {
    MfmaForm f;
    if (!mfma_form(variant, &f)) return false;
    *rows_per_block = 32u * (uint32_t)f.tiles * (uint32_t)f.waves;
    *col_chunk = pick_col_chunk_m(fp4_rows_padded(n), *rows_per_block);
    return true;
}

static HitCtx hit_ctx(const AllPairsArgs& a, bool rect, uint32_t nq, const int32_t* d_group_t, int s1, const void* d_img_q,
                      const void* d_img_t) {
    HitCtx c;
    c.group = a.d_group;
    c.group_t = d_group_t;
    c.out = a.d_pairs;
    c.cap = a.cap;
    c.count = a.d_count;
    c.vs = a.sink;
    c.n = a.n;
    c.nq = nq;
    c.thr_full = 256.0f - 2.0f * (float)a.max_dist;
    c.rect = rect ? 1u : 0u;
    c.acc_start = (float)(64 * s1) - 2.0f * (float)a.max_dist - 1.0f;
    c.img_q = (const uint4*)d_img_q;
    c.img_t = (const uint4*)d_img_t;
    c.db_t = (const uint4*)a.d_db;
    c.db_q = (const uint4*)(rect ? a.d_db_q : a.d_db);
    if (c.db_q == nullptr || c.db_t == nullptr || !g_mfma_queue_packed) c.db_q = c.db_t = nullptr;
    c.max_dist = a.max_dist;
    uint32_t* sel = nullptr;
    c.clk = mfma_select_buffer(a.ctx_id, &sel) == hipSuccess ? reinterpret_cast<unsigned long long*>(sel + kClkWord) : nullptr;
    return c;
}

// One launch of one form. rect: rows = the nq hashes of d_img_q, columns = the a.n hashes of d_img.
template <int T, int NBR, int S1, bool QUEUE = false>
static hipError_t launch_form(const AllPairsArgs& a, const void* d_img, bool rect, const void* d_img_q, uint32_t nq,
                              const int32_t* d_group_t, const uint32_t* d_select, uint32_t select_id, hipStream_t s,
                              bool write_ctx = true) {
    const uint32_t n_pad = fp4_rows_padded(a.n);
    constexpr uint32_t ROWS = 32u * T * 4u;
    const uint32_t nrows = rect ? nq : a.n;
    const uint64_t n_rb = (nrows + ROWS - 1) / ROWS;
    uint64_t chunk;
    if (!rect) {
        chunk = pick_col_chunk_m(n_pad, ROWS);
    } else {  // column chunk sized for the rectangle: enough tiles to fill the chip even when nq is small
        const uint64_t want_cb = (4096 + n_rb - 1) / n_rb;
        chunk = (n_pad + want_cb - 1) / want_cb;
        if (chunk < 256) chunk = 256;
        if (chunk > 4096) chunk = 4096;
        chunk = (chunk + kSuper - 1) / kSuper * kSuper;
        if ((n_pad + chunk - 1) / chunk > 65535u) chunk = round_up((n_pad + 65534u) / 65535u, kSuper);
    }
    const uint64_t n_cb = (n_pad + chunk - 1) / chunk;
    dim3 grid((unsigned)n_rb, (unsigned)(a.world > 1u ? (n_cb + a.world - 1) / a.world : n_cb));  // (own tiles only: see the kernel)
    uint32_t* buf = nullptr;
    hipError_t e = mfma_select_buffer(a.ctx_id, &buf);
    if (e != hipSuccess) return e;
    HitCtx* ctx = reinterpret_cast<HitCtx*>(buf + 16);
    // (the auto variant writes one context for its three launches: they share S1, the only form-dependent field)
    if (write_ctx)
        hipLaunchKernelGGL(k_set_hit_ctx, dim3(1), dim3(1), 0, s, ctx, hit_ctx(a, rect, nq, d_group_t, S1, rect ? d_img_q : d_img, d_img),
                           (uint32_t*)nullptr);
    if (rect)
        hipLaunchKernelGGL((k_allpairs_mfma<T, NBR, S1, true, QUEUE>), grid, dim3(256), 0, s, (const uint4*)d_img, a.n, n_pad,
                           a.max_dist, (uint32_t)chunk, a.rank, a.world, (const uint4*)d_img_q, ctx, d_select,
                           select_id, (uint32_t)(g_mfma_force_sel > 0 ? g_mfma_force_sel : 0));
    else
        hipLaunchKernelGGL((k_allpairs_mfma<T, NBR, S1, false, QUEUE>), grid, dim3(256), 0, s, (const uint4*)d_img, a.n, n_pad,
                           a.max_dist, (uint32_t)chunk, a.rank, a.world, (const uint4*)nullptr, ctx, d_select,
                           select_id, (uint32_t)(g_mfma_force_sel > 0 ? g_mfma_force_sel : 0));
    return hipGetLastError();
}

static hipError_t launch_variant(int variant, const AllPairsArgs& a, const void* d_img, bool rect, const void* d_img_q,
                                 uint32_t nq, const int32_t* d_group_t, const uint32_t* d_select, hipStream_t s,
                                 bool write_ctx = true) {
    switch (variant) {
        case 8: return launch_form<8, 4, 4>(a, d_img, rect, d_img_q, nq, d_group_t, d_select, 8u, s, write_ctx);
        case 9: return launch_form<8, 2, 2>(a, d_img, rect, d_img_q, nq, d_group_t, d_select, 9u, s, write_ctx);
        case 12: return launch_form<4, 4, 2>(a, d_img, rect, d_img_q, nq, d_group_t, d_select, 12u, s, write_ctx);
        case 18: return launch_form<8, 2, 2, true>(a, d_img, rect, d_img_q, nq, d_group_t, d_select, 18u, s, write_ctx);
        default: return hipErrorInvalidValue;
    }
}

// select[0] = form to run, select[1] / select[2] / select[5] = first-stage survivors the probe counted over bits 0..127 /
// 128..255 / 0..63 + 192..255, select[3] = the selection the first stage runs on (0 / 1 / 2), select[4] = the probe's ticket. One buffer per CONTEXT of the library (a context = one stream on one
// device; a group may hold two contexts on one device, whose passes run concurrently on their own streams).
constexpr int kMaxSelect = 16;
static uint32_t* g_select[kMaxSelect] = {};
// The hit context and the select words are device state written in stream order right before the kernels that
// read them: two host threads enqueueing passes at once must not interleave "write context, launch" sequences.
static std::mutex g_launch_mu;

hipError_t mfma_select_buffer(int ctx_id, uint32_t** out) {
    if (ctx_id < 0 || ctx_id >= kMaxSelect) return hipErrorInvalidValue;
    static std::mutex alloc_mu;
    std::lock_guard<std::mutex> lk(alloc_mu);
    if (!g_select[ctx_id]) {
        static_assert(sizeof(HitCtx) <= 192, "hit context does not fit its slot");
        // 24 B of select words, the hit context at +64, the clock telemetry's four accumulators at +768 (kClkWord)
        uint32_t* p = nullptr;
        hipError_t e = hipMalloc((void**)&p, 1024);
        // (hipMemset on device memory does not wait: without the synchronisation it can land on top of the first context
        // that the non-blocking library stream writes)
        if (e == hipSuccess) e = hipMemset(p, 0, 1024);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            if (p) (void)hipFree(p);
            return e;
        }
        g_select[ctx_id] = p;
    }
    *out = g_select[ctx_id];
    return hipSuccess;
}

// Clock telemetry of the all-pairs passes of one context (k_allpairs_mfma): reset enqueues the zeroing on the stream; read
// waits for the stream and returns the accumulators {shader cycles, constant-rate ticks, sampled workgroups, 0}.
hipError_t mfma_clock_reset(int ctx_id, hipStream_t s) {
    uint32_t* sel = nullptr;
    hipError_t e = mfma_select_buffer(ctx_id, &sel);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_clk_reset, dim3(1), dim3(1), 0, s, reinterpret_cast<unsigned long long*>(sel + kClkWord));
    return hipGetLastError();
}

hipError_t mfma_clock_read(int ctx_id, hipStream_t s, unsigned long long out[4]) {
    uint32_t* sel = nullptr;
    hipError_t e = mfma_select_buffer(ctx_id, &sel);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(out, sel + kClkWord, 32, hipMemcpyDeviceToHost, s);
    return e == hipSuccess ? hipStreamSynchronize(s) : e;
}

void mfma_release() {
    for (uint32_t*& p : g_select) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
}

// Probe, decide on the device, launch both candidate forms: the one the probe did not choose returns at once.
// No host synchronisation. The DB is replicated and the probe is deterministic, so every rank of a multi-GPU
// pass picks the same form (the tile partition depends on it).
static hipError_t launch_auto(const AllPairsArgs& a, const void* d_img, bool rect, const void* d_img_q, uint32_t nq,
                              const int32_t* d_group_t, hipStream_t s) {
    uint32_t* sel = nullptr;
    hipError_t e = mfma_select_buffer(a.ctx_id, &sel);
    if (e != hipSuccess) return e;
    // one launch writes the hit context of all three forms (they share S1 = 2) and clears the probe's words; the probe's last
    // workgroup decides. Per pass: context, probe, three forms -- five launches where there were nine.
    hipLaunchKernelGGL(k_set_hit_ctx, dim3(1), dim3(1), 0, s, reinterpret_cast<HitCtx*>(sel + 16),
                       hit_ctx(a, rect, nq, d_group_t, 2, rect ? d_img_q : d_img, d_img), sel);
    const uint32_t nrows = rect ? nq : a.n;
    const uint32_t rows = nrows < kProbeRows ? nrows : kProbeRows, cols = a.n < kProbeCols ? a.n : kProbeCols;
    // (the pair queue keeps (column << 1 | half) in 32 bits)
    const uint32_t mid = fp4_rows_padded(a.n) < (1u << 31) ? g_mfma_auto_mid : 0u;
    const ProbeRule rule = {(uint64_t)rows * cols, 8192u, 9u, mid, 12u, 0.01f * (float)g_mfma_auto_mid_max_x100, g_mfma_force_sel};
    hipLaunchKernelGGL(k_prefilter_probe, dim3((rows + 255u) / 256u, (cols + kProbeColsPerWg - 1u) / kProbeColsPerWg), dim3(256), 0, s,
                       (const uint4*)(rect ? d_img_q : d_img), nrows, (const uint4*)d_img, a.n, a.max_dist, sel, rule);
    if (a.sync_decide) {
        uint32_t form = 0;
        e = hipMemcpyAsync(&form, sel, 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        if (form != 9u && form != 12u && form != mid) return hipErrorUnknown;  // (the probe writes one of its rule's three ids)
        return launch_variant((int)form, a, d_img, rect, d_img_q, nq, d_group_t, sel, s, false);
    }
    e = launch_variant(9, a, d_img, rect, d_img_q, nq, d_group_t, sel, s, false);
    if (e != hipSuccess) return e;
    if (mid) {
        e = launch_variant((int)mid, a, d_img, rect, d_img_q, nq, d_group_t, sel, s, false);
        if (e != hipSuccess) return e;
    }
    return launch_variant(12, a, d_img, rect, d_img_q, nq, d_group_t, sel, s, false);
}

static int effective_variant(int variant, uint32_t max_dist, uint32_t n) {
    if (variant == 18 && fp4_rows_padded(n) >= (1u << 31)) variant = 12;  // the pair queue keeps (column << 1 | half) in 32 bits
    // the 128-bit first stage needs 128 - 2*max_dist > 0
    if (max_dist >= 64u && (variant == 9 || variant == 12 || variant == 13 || variant == 18)) return 8;
    return variant;
}

hipError_t launch_cross_mfma(const AllPairsArgs& a, const void* d_img_q, uint32_t nq, const void* d_img_t,
                             const int32_t* d_group_t, hipStream_t s) {
    if (nq == 0 || a.n == 0) return hipSuccess;
    std::lock_guard<std::mutex> lk(g_launch_mu);
    if (a.max_dist >= 128u) return hipErrorInvalidValue;  // sign trick needs a positive threshold
    const int v = effective_variant(a.variant, a.max_dist, a.n);
    if (v == 13) return launch_auto(a, d_img_t, true, d_img_q, nq, d_group_t, s);
    return launch_variant(v, a, d_img_t, true, d_img_q, nq, d_group_t, nullptr, s);
}

hipError_t launch_allpairs_mfma(const AllPairsArgs& a_in, const void* d_img, hipStream_t s) {
    if (a_in.n < 2) return hipSuccess;
    AllPairsArgs a = a_in;
    // The sign trick of max16_bits needs a positive threshold: 256 - 2*max_dist > 0. Larger
    // tolerances (never used by the reference, which fixes 31) go to the popcount kernel, which
    // shares the tile geometry class.
    if (a.max_dist >= 128u) {
        if (!a.d_db || a.sink.set) return hipErrorInvalidValue;
        a.variant = 0;
        return launch_allpairs(a, s);
    }
    const int v = effective_variant(a.variant, a.max_dist, a.n);
    std::lock_guard<std::mutex> lk(g_launch_mu);
    if (v == 13) return launch_auto(a, d_img, false, nullptr, 0u, nullptr, s);
    return launch_variant(v, a, d_img, false, nullptr, 0u, nullptr, nullptr, s);
}

}  // namespace hvd
