// k_pdq.hip -- PDQ frame hashing for gfx950 (MI355X).
//
// Replaces the per-frame arithmetic behind vpdq.VideoHasher.hash_frame
// (reference vpdqpy/vpdqpy.py:113-119): luma -> [Jarosz blur -> decimate] -> 64x64
// -> quality metric -> 16x16 DCT -> median -> 256 bits, laid out as
// db/DedupeDB.py:535-559 describes (bit k=i*16+j at byte k>>3, bit k&7).
//
// Bit-exactness contract (oracle/hvd_oracle.c): every float op is a separately
// rounded binary32 op in the oracle's order. All arithmetic that must match uses
// __fmul_rn/__fadd_rn/__fsub_rn/__fdiv_rn, which hipcc never contracts into FMA,
// and the DCT matrix is computed on the host in double and uploaded.
// The f32 MFMA is an fmaf chain (one rounding per product+add) and therefore NOT
// bit-identical to the reference's mul-then-add on x86-64; the DCT runs on the VALU.
//
// k_pdq_hash64: one wave64 per frame, 4 frames per workgroup, persistent grid.
//   stage 0  lane j loads column j of the 64x64 frame (64 values in VGPRs)
//   quality  vertical gradients in-lane, horizontal ones via the neighbour lane
//   stage 1  T = D*A: lane j owns column j, D[i][k] is wave-uniform -> scalar
//            loads, SGPR operands; 4 independent k-sequential chains per pass
//   stage 2  B = T*D^T through LDS (T: 16x64, padded), lane l -> (i0=l>>4, j=l&15),
//            4 outputs per lane; D rows come from a padded LDS copy
//   median   radix select of the 128th smallest of 256 keys with wave ballots
//   bits     ballot(B > median): lane l, output r is hash bit l + 64 r
#include <hip/hip_runtime.h>

// HVD_ABL_NOSTATE / HVD_ABL_NOFETCH / HVD_ABL_NOD / HVD_ABL_NOLUMA / HVD_ABL_NOSTATELOAD builds are timing-only ABLATIONS of the down-sampler that produce
// WRONG RESULTS. They may not come out of the product source with a single -D:
#if (defined(HVD_ABL_NOSTATE) || defined(HVD_ABL_NOFETCH) || defined(HVD_ABL_NOD) || defined(HVD_ABL_NOLUMA) || defined(HVD_ABL_NOSTATELOAD)) && !defined(HVD_DEV_ABLATION)
#error "HVD_ABL_* are developer ablation builds (wrong results): add -DHVD_DEV_ABLATION to confirm"
#endif
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "hvd_kernels.h"

namespace {

constexpr int kWaves = 4;       // frames in flight per workgroup
constexpr int kLd = 68;         // padded LDS row stride (floats): 272 B, 16-B aligned, bank-skewed

struct alignas(16) PdqLds {
    float T[kWaves][16][kLd];
    float D[16][kLd];
    float luma_lut[256];  // luma_gray(g) for every byte value
};

__device__ __forceinline__ float luma_gray(uint32_t g) {
    const float v = (float)g;
    float y = __fmul_rn(0.299f, v);
    y = __fadd_rn(y, __fmul_rn(0.587f, v));
    y = __fadd_rn(y, __fmul_rn(0.114f, v));
    return y;
}

// |(int)(((u - v) * 100) / 255)| (pdqhashing.cpp quality metric) as a non-negative integer-valued
// float, WITHOUT the IEEE division (10+ VALU ops). The multiplier is the float just BELOW 1/255
// (RN(1/255) = 0x1.010102p-8 lies above the true value), so q = |x| * c stays below the true
// quotient even after its own rounding: trunc(q) is floor(|x|/255) or one less. The remainder
// r = |x| - 255*m is exact in one fma (|x| and 255*m are multiples of ulp(|x|) and close) and
// r >= 255 says when to add one. Equality with (int)(x / 255.0f) is checked for EVERY float
// |x| <= 26000 by tests/tools/check_div255.c (2.4e9 values, 0 mismatches); |x| <= 25500.01 here
// because luma and its box-filter averages never exceed 255.0001. The fma is this kernel's own
// exact-arithmetic device, not a contraction of reference arithmetic.
// The term is m + (r >= 255): the caller accumulates the m's as floats (exact: integers far below
// 2^24) and the corrections as an integer count (v_cmp + add-with-carry).
__device__ __forceinline__ void grad_term(float u, float v, float& acc_m, int& acc_c) {
    const float ax = fabsf(__fmul_rn(__fsub_rn(u, v), 100.0f));
    const float m = truncf(__fmul_rn(ax, 0x1.0101p-8f));
    const float r = __fmaf_rn(-255.0f, m, ax);
    acc_m += m;
    acc_c += (r >= 255.0f) ? 1 : 0;
}

// The same term for GRAY BYTE input in one multiply: there the operands are luma_gray(g) of a byte g, so (u, v) takes
// only 256 x 256 values, and for every one of them trunc(|u - v| * RN(100/255)) equals the reference's
// |(int)(((u - v) * 100) / 255)| -- checked exhaustively (tests/test_oracle.py::test_quality_term_gray_shortcut_is_exact
// on the host, test_k1_quality_all_byte_pairs on the GPU). 4 VALU ops per term instead of 8; the quality metric was a
// quarter of this kernel's instructions (profiles/r01_pmc_k1.txt). Float frames (the down-sampler's output) keep the
// general form above.
__device__ __forceinline__ void grad_term_gray(float u, float v, int& acc) {
    // (int)x IS the truncation (v_cvt_i32_f32 rounds toward zero); an explicit truncf in front of it cost one more VALU
    // instruction per term, 127 per frame (round 3)
    acc += (int)__fmul_rn(fabsf(__fsub_rn(u, v)), 0x1.919192p-2f /* RN(100/255) = 0x3EC8C8C9 */);
}

// Lane l reads lane l+1 of the whole 64-lane wave (lane 63 reads 0 and is ignored by callers): the DPP
// wave_shl:1 control of the GFX9 family, which folds into the consuming VALU instruction instead
// of a trip through the LDS crossbar (ds_bpermute).
__device__ __forceinline__ float wave_next_lane(float v) {
    const int x = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}

__device__ __forceinline__ float wave_sum_f32(float v) {  // exact: integer-valued, far below 2^24
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The DCT matrix as compile-time constants (scripts/gen_dct_table.py). This table is authoritative: hvd_init does not
// consult the host's libm (tests/ compare the table with hvd_dct_matrix_libm() and with the oracle).
constexpr uint32_t kDctBits[16][64] = {
#include "dct_table.inc"
};
__device__ __forceinline__ constexpr float dct_lit(int i, int k) { return __builtin_bit_cast(float, kDctBits[i][k]); }

__device__ __forceinline__ void wave_lds_handover() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// KIND 0: uint8 gray 64x64 frames. KIND 1: float 64x64 buffers (down-sampler output).
// DLDS: where stage 1 takes D[i][k] from. 0: scalar loads (SGPR operands: v_mul_f32 s,v issues at half rate,
// profiles/r01_ubench_valu.txt). 1: LDS broadcast reads (VGPR operands, full-rate v_mul, but an LDS read per 4
// products). 2: 32-bit LITERALS in the instruction stream -- the matrix is a constant of the algorithm, so stage 1 is
// unrolled completely (16 x 64 multiply-adds, ~20 KB of code) and every multiply carries its coefficient: no operand
// fetch at all, full-rate issue.
// Work distribution (round 3): a workgroup takes chunk blockIdx.x (`chunk` consecutive groups of 4 frames) statically and
// every further chunk from a counter in device memory, so that no workgroup is left with a trip more than its neighbours
// while their SIMDs idle (at 400 k frames the static stride cost 16 %). One atomic per trip, ISSUED at the top of the trip
// and CONSUMED at its end: same-address device atomics serialise at ~8 ns each, and a thousand workgroups draw at once when
// a launch starts -- which is why launches below 64 k frames keep the static stride (work == nullptr): measured, the draws cost
// 10 k frames 52 -> 65 us, while 400 k frames gain 12 % (profiles/r03_k1_grid.txt). Every trip draws exactly once, so the draw that returns (trips - 1) is the last of the launch: its
// workgroup zeroes the counter for the next launch that is handed this slot -- no exit counter.
// PREF (gray bytes, static stride only -- the launches below 64 k frames): the wave's NEXT frame is fetched while it works
// on this one, as four 16-byte loads per lane (16 VGPRs in flight for a whole frame's time), handed to the lanes through
// the wave's own T area (4 096 of its 4 352 bytes; T is written only after the last byte has been read) -- instead of 64
// byte loads per lane whose latency every frame started with (the fma kernel has worked this way since round 2).
#ifdef HVD_K1_WAVES  // A/B builds: force the strict hash kernel to this many waves per SIMD (profiles/r04_k1_grid.txt)
#define HVD_K1_OCC __attribute__((amdgpu_waves_per_eu(HVD_K1_WAVES, HVD_K1_WAVES)))
#else
#define HVD_K1_OCC
#endif
template <int KIND, int DLDS, int LUT, bool PREF = false>
__global__ __launch_bounds__(256) HVD_K1_OCC void k_pdq_hash64(const void* __restrict__ in, long long n,
                                                    const float* __restrict__ dct, uint8_t* __restrict__ hashes,
                                                    int32_t* __restrict__ quality, unsigned int* __restrict__ work,
                                                    int chunk) {
    __shared__ PdqLds lds;
    __shared__ unsigned int next_chunk;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // Padded LDS copy of the DCT matrix for stage 2 (once per workgroup).
    for (int e = threadIdx.x; e < 16 * 64; e += 256) lds.D[e >> 6][e & 63] = dct[e];
    lds.luma_lut[threadIdx.x] = luma_gray(threadIdx.x);
    __syncthreads();

    const long long groups = (n + kWaves - 1) / kWaves;
    const long long nchunks = (groups + chunk - 1) / chunk;
    uint4 nb0 = make_uint4(0, 0, 0, 0), nb1 = nb0, nb2 = nb0, nb3 = nb0;  // PREF: the next frame's bytes, in flight
    if (PREF) {
        const long long f0 = (long long)blockIdx.x * kWaves + wave;
        if ((long long)blockIdx.x < groups && f0 < n) {
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + f0 * 4096);
            nb0 = src[lane]; nb1 = src[64 + lane]; nb2 = src[128 + lane]; nb3 = src[192 + lane];
        }
    }
    for (long long ck = blockIdx.x; ck < nchunks;) {
        unsigned int drawn = 0;
        if (work != nullptr && threadIdx.x == 0) drawn = atomicAdd(&work[0], 1u);  // consumed behind this trip's last group
      for (long long g = ck * chunk; g < groups && g < (ck + 1) * chunk; ++g) {
        const long long f = g * kWaves + wave;
        const bool valid = f < n;  // wave-uniform

        if (valid) {
            // ---- stage 0: column `lane` of the frame -------------------------------
            float a[64];
            if (KIND == 0 && PREF) {
                uint4* dstb = reinterpret_cast<uint4*>(&lds.T[wave][0][0]);
                dstb[lane] = nb0; dstb[64 + lane] = nb1; dstb[128 + lane] = nb2; dstb[192 + lane] = nb3;
                const long long fn = (g + gridDim.x) * kWaves + wave;  // (static stride, one group per trip)
                if (g + gridDim.x < groups && fn < n) {
                    const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + fn * 4096);
                    nb0 = src[lane]; nb1 = src[64 + lane]; nb2 = src[128 + lane]; nb3 = src[192 + lane];
                }
                wave_lds_handover();
                const uint8_t* srcb = reinterpret_cast<const uint8_t*>(&lds.T[wave][0][0]) + lane;
#pragma unroll
                for (int k = 0; k < 64; ++k) a[k] = lds.luma_lut[srcb[k * 64]];
                wave_lds_handover();  // every byte has been read before stage 1 writes T over them
            } else if (KIND == 0) {
                const uint8_t* src = reinterpret_cast<const uint8_t*>(in) + f * 4096 + lane;
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    if (LUT == 0) {
                        a[k] = luma_gray(src[k * 64]);
                    } else {
                        a[k] = lds.luma_lut[src[k * 64]];
                        if (LUT == 2 && (k & 15) == 15) __builtin_amdgcn_sched_barrier(0);  // bound the loads in flight
                    }
                }
            } else {
                const float* src = reinterpret_cast<const float*>(in) + f * 4096 + lane;
#pragma unroll
                for (int k = 0; k < 64; ++k) a[k] = src[k * 64];
            }

            // ---- quality -----------------------------------------------------------
            int gsum;
            if (KIND == 0) {
                int qs = 0, qh = 0;
#pragma unroll
                for (int k = 0; k < 63; ++k) grad_term_gray(a[k], a[k + 1], qs);
#pragma unroll
                for (int k = 0; k < 64; ++k) grad_term_gray(a[k], wave_next_lane(a[k]), qh);
                if (lane < 63) qs += qh;  // column 63 has no right neighbour
                gsum = (int)wave_sum_f32((float)qs);
            } else {
                float gs = 0.0f, gh = 0.0f;
                int cs_ = 0, ch_ = 0;
#pragma unroll
                for (int k = 0; k < 63; ++k) grad_term(a[k], a[k + 1], gs, cs_);
#pragma unroll
                for (int k = 0; k < 64; ++k) grad_term(a[k], wave_next_lane(a[k]), gh, ch_);
                if (lane < 63) {  // column 63 has no right neighbour
                    gs += gh;
                    cs_ += ch_;
                }
                gsum = (int)wave_sum_f32(gs + (float)cs_);
            }
            int qual = gsum / 90;
            qual = qual > 100 ? 100 : qual;

            // ---- stage 1: T[i][lane] = sum_k D[i][k] * a[k], k ascending ------------
            if (DLDS == 2) {
#pragma unroll
                for (int i0 = 0; i0 < 16; i0 += 4) {
                    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
                    for (int k = 0; k < 64; ++k) {
                        s0 = __fadd_rn(s0, __fmul_rn(dct_lit(i0 + 0, k), a[k]));
                        s1 = __fadd_rn(s1, __fmul_rn(dct_lit(i0 + 1, k), a[k]));
                        s2 = __fadd_rn(s2, __fmul_rn(dct_lit(i0 + 2, k), a[k]));
                        s3 = __fadd_rn(s3, __fmul_rn(dct_lit(i0 + 3, k), a[k]));
                    }
                    lds.T[wave][i0 + 0][lane] = s0;
                    lds.T[wave][i0 + 1][lane] = s1;
                    lds.T[wave][i0 + 2][lane] = s2;
                    lds.T[wave][i0 + 3][lane] = s3;
                }
            } else
#pragma unroll 1
            for (int i0 = 0; i0 < 16; i0 += 4) {
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
                if (DLDS == 1) {
#pragma unroll
                    for (int k4 = 0; k4 < 16; ++k4) {  // same address in every lane: LDS broadcast
                        const float4 e0 = *reinterpret_cast<const float4*>(&lds.D[i0 + 0][4 * k4]);
                        const float4 e1 = *reinterpret_cast<const float4*>(&lds.D[i0 + 1][4 * k4]);
                        const float4 e2 = *reinterpret_cast<const float4*>(&lds.D[i0 + 2][4 * k4]);
                        const float4 e3 = *reinterpret_cast<const float4*>(&lds.D[i0 + 3][4 * k4]);
                        const float x0 = a[4 * k4], x1 = a[4 * k4 + 1], x2 = a[4 * k4 + 2], x3 = a[4 * k4 + 3];
                        s0 = __fadd_rn(s0, __fmul_rn(e0.x, x0)); s1 = __fadd_rn(s1, __fmul_rn(e1.x, x0));
                        s2 = __fadd_rn(s2, __fmul_rn(e2.x, x0)); s3 = __fadd_rn(s3, __fmul_rn(e3.x, x0));
                        s0 = __fadd_rn(s0, __fmul_rn(e0.y, x1)); s1 = __fadd_rn(s1, __fmul_rn(e1.y, x1));
                        s2 = __fadd_rn(s2, __fmul_rn(e2.y, x1)); s3 = __fadd_rn(s3, __fmul_rn(e3.y, x1));
                        s0 = __fadd_rn(s0, __fmul_rn(e0.z, x2)); s1 = __fadd_rn(s1, __fmul_rn(e1.z, x2));
                        s2 = __fadd_rn(s2, __fmul_rn(e2.z, x2)); s3 = __fadd_rn(s3, __fmul_rn(e3.z, x2));
                        s0 = __fadd_rn(s0, __fmul_rn(e0.w, x3)); s1 = __fadd_rn(s1, __fmul_rn(e1.w, x3));
                        s2 = __fadd_rn(s2, __fmul_rn(e2.w, x3)); s3 = __fadd_rn(s3, __fmul_rn(e3.w, x3));
                    }
                } else {
                    const float* d0 = dct + (i0 + 0) * 64;  // wave-uniform -> s_load
                    const float* d1 = dct + (i0 + 1) * 64;
                    const float* d2 = dct + (i0 + 2) * 64;
                    const float* d3 = dct + (i0 + 3) * 64;
#pragma unroll
                    for (int k = 0; k < 64; ++k) {
                        s0 = __fadd_rn(s0, __fmul_rn(d0[k], a[k]));
                        s1 = __fadd_rn(s1, __fmul_rn(d1[k], a[k]));
                        s2 = __fadd_rn(s2, __fmul_rn(d2[k], a[k]));
                        s3 = __fadd_rn(s3, __fmul_rn(d3[k], a[k]));
                    }
                }
                lds.T[wave][i0 + 0][lane] = s0;
                lds.T[wave][i0 + 1][lane] = s1;
                lds.T[wave][i0 + 2][lane] = s2;
                lds.T[wave][i0 + 3][lane] = s3;
            }
            if (lane == 0) quality[f] = qual;
        }
        // T[wave] is private to this wave: wave-scope ordering is all the hand-over needs (LDS operations of one wave
        // execute in order), so the four waves of a workgroup -- four independent frames -- never wait for each other
        // (round 4; two workgroup barriers per frame cost ~0.35 of the launch at 10 k frames)
        wave_lds_handover();

        if (valid) {
            // ---- stage 2: B[i][j] = sum_k T[i][k] * D[j][k], k ascending ------------
            const int j = lane & 15, i0 = lane >> 4;
            float b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k4 = 0; k4 < 16; ++k4) {
                const float4 dv = *reinterpret_cast<const float4*>(&lds.D[j][4 * k4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 tv = *reinterpret_cast<const float4*>(&lds.T[wave][i0 + 4 * r][4 * k4]);
                    b[r] = __fadd_rn(b[r], __fmul_rn(tv.x, dv.x));
                    b[r] = __fadd_rn(b[r], __fmul_rn(tv.y, dv.y));
                    b[r] = __fadd_rn(b[r], __fmul_rn(tv.z, dv.z));
                    b[r] = __fadd_rn(b[r], __fmul_rn(tv.w, dv.w));
                }
            }

            // ---- median: 128th smallest of the 256 coefficients (Torben's result) ---
            uint32_t key[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t u = __float_as_uint(b[r]);
                key[r] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving
            }
            uint32_t prefix = 0, mask = 0;
            int kth = 128, remaining = 256;
#pragma unroll 1
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t bsel = 1u << bit;
                const uint32_t m2 = mask | bsel;
                int cnt0 = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) cnt0 += __popcll(__ballot((key[r] & m2) == prefix));
                if (kth > cnt0) {
                    kth -= cnt0;
                    remaining -= cnt0;
                    prefix |= bsel;
                } else {
                    remaining = cnt0;
                }
                mask = m2;
                if (remaining == 1) break;  // a single key carries this prefix: it is the median
            }
            if (mask != 0xFFFFFFFFu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned long long bm = __ballot((key[r] & mask) == prefix);
                    if (bm) prefix = __builtin_amdgcn_readlane(key[r], (int)__builtin_ctzll(bm));
                }
            }
            const uint32_t mu = (prefix & 0x80000000u) ? (prefix ^ 0x80000000u) : ~prefix;
            const float med = __uint_as_float(mu);

            // ---- bits: lane l, output r is coefficient (i0+4r, j) = bit l + 64 r ----
            unsigned long long m[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = __ballot(b[r] > med);
            if (lane < 4) {
                const unsigned long long w = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
                reinterpret_cast<unsigned long long*>(hashes)[f * 4 + lane] = w;
            }
        }
        wave_lds_handover();  // T[wave] is rewritten by this wave's next trip
      }
        if (work == nullptr) {  // short launches stride statically: a thousand workgroups drawing at once wait on each other
            ck += gridDim.x;
            continue;
        }
        if (threadIdx.x == 0) {
            next_chunk = gridDim.x + drawn;
            if ((long long)drawn == nchunks - 1) work[0] = 0u;  // the launch's last draw: nobody will touch the counter again
        }
        __syncthreads();
        ck = next_chunk;
        __syncthreads();  // (everyone has read next_chunk before thread 0 overwrites it at the end of the next trip)
    }
}

// ---------------------------------------------------------------------------
// k_pdq_hash64_fma: the same frame hash with the DCT on the matrix cores ("fma" DCT mode).
//
// v_mfma_f32_16x16x4_f32 is bit for bit a k-ordered fmaf chain (one rounding per multiply-add),
// i.e. exactly what upstream's `sumk += D*A` becomes when the compiler contracts it -- the
// numerics of its arm64 wheels, NOT of the x86-64 ones (those are the default "strict" mode
// above). The mode is opt-in (hvd_set_pdq_dct_mode) and is held to the same standard: bit-exact
// against the oracle's fma mode.
//   stage 1  T(16x64) = D(16x64) * A(64x64): per 16-column block nb, 16 MFMAs over k;
//            A operand = D[i = lane&15][k = 4ks + (lane>>4)] (constant, 16 VGPRs per wave),
//            B operand = luma(row 4ks + (lane>>4), column 16nb + (lane&15))
//   stage 2  B(16x16) = T * D^T: A operand = T[i = lane&15][k] (through LDS), B operand = the same
//            D fragments; output B[i = 4(lane>>4) + r][j = lane&15]
// Quality, median and bit extraction are shared with the strict kernel (the quality needs the
// column-per-lane view, so the frame bytes are read in both layouts; they are L1-hot).
typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256, 3) void k_pdq_hash64_fma(const void* __restrict__ in, long long n,
                                                        const float* __restrict__ dct, uint8_t* __restrict__ hashes,
                                                        int32_t* __restrict__ quality) {
    __shared__ PdqLds lds;
    // KIND 0: the wave's frame goes through LDS: four 16-byte loads per lane, issued ONE FRAME AHEAD (16 VGPRs),
    // instead of 128 byte loads whose latency each wave sat out twice per frame (PMC: 44 % of wave time in
    // s_waitcnt with 3 waves per SIMD).
    __shared__ __attribute__((aligned(16))) uint8_t fbytes[KIND == 0 ? kWaves : 1][KIND == 0 ? 4096 : 16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g4 = lane >> 4, c16 = lane & 15;

    lds.luma_lut[threadIdx.x] = luma_gray(threadIdx.x);
    float dfrag[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) dfrag[ks] = dct[c16 * 64 + 4 * ks + g4];
    __syncthreads();

    const long long groups = (n + kWaves - 1) / kWaves;
    uint4 nb0 = make_uint4(0, 0, 0, 0), nb1 = nb0, nb2 = nb0, nb3 = nb0;  // the next frame's bytes, in flight
    if (KIND == 0) {
        const long long f0 = (long long)blockIdx.x * kWaves + wave;
        if ((long long)blockIdx.x < groups && f0 < n) {
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + f0 * 4096);
            nb0 = src[lane]; nb1 = src[64 + lane]; nb2 = src[128 + lane]; nb3 = src[192 + lane];
        }
    }
    for (long long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long long f = g * kWaves + wave;
        const bool valid = f < n;  // wave-uniform
        float b[4] = {0.0f, 0.0f, 0.0f, 0.0f};

        if (valid) {
            if (KIND == 0) {
                uint4* dstb = reinterpret_cast<uint4*>(&fbytes[wave][0]);
                dstb[lane] = nb0; dstb[64 + lane] = nb1; dstb[128 + lane] = nb2; dstb[192 + lane] = nb3;
                const long long fn = (g + gridDim.x) * kWaves + wave;
                if (g + gridDim.x < groups && fn < n) {
                    const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(in) + fn * 4096);
                    nb0 = src[lane]; nb1 = src[64 + lane]; nb2 = src[128 + lane]; nb3 = src[192 + lane];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            // ---- quality: column `lane` of the frame, as in the strict kernel ---------------------
            {
                float a[64];
                if (KIND == 0) {
                    const uint8_t* src = &fbytes[wave][lane];
#pragma unroll
                    for (int k = 0; k < 64; ++k) a[k] = lds.luma_lut[src[k * 64]];
                } else {
                    const float* src = reinterpret_cast<const float*>(in) + f * 4096 + lane;
#pragma unroll
                    for (int k = 0; k < 64; ++k) a[k] = src[k * 64];
                }
                int gsum;
                if (KIND == 0) {  // gray bytes: the one-multiply form (grad_term_gray)
                    int qs = 0, qh = 0;
#pragma unroll
                    for (int k = 0; k < 63; ++k) grad_term_gray(a[k], a[k + 1], qs);
#pragma unroll
                    for (int k = 0; k < 64; ++k) grad_term_gray(a[k], wave_next_lane(a[k]), qh);
                    if (lane < 63) qs += qh;
                    gsum = (int)wave_sum_f32((float)qs);
                } else {
                    float gs = 0.0f, gh = 0.0f;
                    int cs_ = 0, ch_ = 0;
#pragma unroll
                    for (int k = 0; k < 63; ++k) grad_term(a[k], a[k + 1], gs, cs_);
#pragma unroll
                    for (int k = 0; k < 64; ++k) grad_term(a[k], wave_next_lane(a[k]), gh, ch_);
                    if (lane < 63) {
                        gs += gh;
                        cs_ += ch_;
                    }
                    gsum = (int)wave_sum_f32(gs + (float)cs_);
                }
                int qual = gsum / 90;
                qual = qual > 100 ? 100 : qual;
                if (lane == 0) quality[f] = qual;
            }

            // ---- stage 1 on the matrix cores --------------------------------------------------------
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float af[16];
                if (KIND == 0) {
                    const uint8_t* src = &fbytes[wave][g4 * 64 + 16 * nb + c16];
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) af[ks] = lds.luma_lut[src[ks * 256]];
                } else {
                    const float* src = reinterpret_cast<const float*>(in) + f * 4096 + g4 * 64 + 16 * nb + c16;
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) af[ks] = src[ks * 256];
                }
                v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dfrag[ks], af[ks], acc, 0, 0, 0);
                // C/D layout: column = lane&15, row = 4*(lane>>4) + r  ->  T[i = 4 g4 + r][j = 16 nb + c16]
#pragma unroll
                for (int r = 0; r < 4; ++r) lds.T[wave][4 * g4 + r][16 * nb + c16] = acc[r];
            }
        }
        __syncthreads();

        if (valid) {
            // ---- stage 2 on the matrix cores: A operand T[i = c16][k = 4ks + g4] ----------------------
            v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds.T[wave][c16][4 * ks + g4], dfrag[ks], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = acc[r];  // B[i = 4 g4 + r][j = c16]

            // ---- median (same radix select as the strict kernel) ---------------------------------------
            uint32_t key[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t u = __float_as_uint(b[r]);
                key[r] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            }
            uint32_t prefix = 0, mask = 0;
            int kth = 128, remaining = 256;
#pragma unroll 1
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t bsel = 1u << bit;
                const uint32_t m2 = mask | bsel;
                int cnt0 = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) cnt0 += __popcll(__ballot((key[r] & m2) == prefix));
                if (kth > cnt0) {
                    kth -= cnt0;
                    remaining -= cnt0;
                    prefix |= bsel;
                } else {
                    remaining = cnt0;
                }
                mask = m2;
                if (remaining == 1) break;
            }
            if (mask != 0xFFFFFFFFu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned long long bm = __ballot((key[r] & mask) == prefix);
                    if (bm) prefix = __builtin_amdgcn_readlane(key[r], (int)__builtin_ctzll(bm));
                }
            }
            const uint32_t mu = (prefix & 0x80000000u) ? (prefix ^ 0x80000000u) : ~prefix;
            const float med = __uint_as_float(mu);

            // ---- bits: lane (g4, c16), output r is coefficient (4 g4 + r, c16) = hash bit 64 g4 + 16 r + c16:
            //      64-bit word w of the hash takes 16 bits from each of the four ballots
            unsigned long long m[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = __ballot(b[r] > med);
            if (lane < 4) {
                const int sh = 16 * lane;
                const unsigned long long w = ((m[0] >> sh) & 0xFFFFull) | (((m[1] >> sh) & 0xFFFFull) << 16) |
                                             (((m[2] >> sh) & 0xFFFFull) << 32) | (((m[3] >> sh) & 0xFFFFull) << 48);
                reinterpret_cast<unsigned long long*>(hashes)[f * 4 + lane] = w;
            }
        }
        __syncthreads();
    }
}

// rgb24 64x64 frames -> float luma (no blur: upstream's 64x64 shortcut).
__global__ __launch_bounds__(256) void k_luma64_rgb(const uint8_t* __restrict__ rgb, long long npix,
                                                    float* __restrict__ out) {
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long long)gridDim.x * 256) {
        const float r = (float)rgb[3 * p], g = (float)rgb[3 * p + 1], b = (float)rgb[3 * p + 2];
        float y = __fmul_rn(0.299f, r);
        y = __fadd_rn(y, __fmul_rn(0.587f, g));
        y = __fadd_rn(y, __fmul_rn(0.114f, b));
        out[p] = y;
    }
}


// ---------------------------------------------------------------------------
// Down-sampler for frames that are not already 64x64 (the reference feeds 512x512
// rgb24, vpdqpy/vpdqpy.py:90-95,113): luma, 2 x (box along rows, box along
// columns) with upstream's sequential running-sum box filter, then decimation.
//
// The running sum makes every 1-D filter a *sequential* float recurrence, so the
// parallelism is across lines, one lane per line. k_box_scan_T filters the lines
// of a row-major [lines][len] image and writes the result TRANSPOSED ([len][lines]),
// so that (a) the input is staged through an LDS ring with coalesced loads, (b) the
// stores are coalesced across lanes, and (c) four launches of the same kernel give
// rows, cols, rows, cols. Passes 3 and 4 only emit the 64 sample positions the
// decimation keeps (the recurrence still runs over every element).
constexpr int kTW = 32;           // columns per staged tile
constexpr int kRing = 2 * kTW;    // LDS ring (window <= 32 looks back at most one tile)

template <int SRC>  // 0: float, 1: gray u8, 3: rgb24 (luma fused into the load)
__global__ __launch_bounds__(64) void k_box_scan_T(const void* __restrict__ in, float* __restrict__ out, int lines,
                                                   int len, int win, int nsel, long long in_frame_stride,
                                                   long long out_frame_stride) {
    __shared__ float ring[64][kRing + 1];
    const int lane = threadIdx.x;
    const int line0 = blockIdx.x * 64;
    const long long frame = blockIdx.y;
    const int my_line = line0 + lane;
    const int half = (win + 2) / 2;
    const int steps = len + half - 1;
    const int out_lines = lines;  // transposed output: [kept positions][lines]
    float* dst = out + frame * out_frame_stride;

    float sum = 0.0f;
    int cur = 0;
    int next_j = 0;
    int next_sel = nsel ? (int)(((0 + 0.5) * len) / 64) : 0;

    for (int s = 0; s < steps; ++s) {
        if (s < len && (s % kTW) == 0) {
            // stage columns [s, s+kTW) of the 64 lines into ring slot (s/kTW)&1
            __syncthreads();
            const int c = lane & (kTW - 1);
            const int col = s + c;
#pragma unroll 4
            for (int rr = lane / kTW; rr < 64; rr += 64 / kTW) {
                const int ln = line0 + rr;
                float v = 0.0f;
                if (ln < lines && col < len) {
                    const long long e = (long long)ln * len + col;
                    if (SRC == 0) {
                        v = reinterpret_cast<const float*>(in)[frame * in_frame_stride + e];
                    } else if (SRC == 1) {
                        v = luma_gray(reinterpret_cast<const uint8_t*>(in)[frame * in_frame_stride + e]);
                    } else {
                        const uint8_t* p = reinterpret_cast<const uint8_t*>(in) + frame * in_frame_stride + 3 * e;
                        const float r = (float)p[0], g = (float)p[1], b = (float)p[2];
                        v = __fmul_rn(0.299f, r);
                        v = __fadd_rn(v, __fmul_rn(0.587f, g));
                        v = __fadd_rn(v, __fmul_rn(0.114f, b));
                    }
                }
                ring[rr][col & (kRing - 1)] = v;
            }
            __syncthreads();
        }
        if (s < len) {
            sum = __fadd_rn(sum, ring[lane][s & (kRing - 1)]);
            if (s < win) ++cur;
        }
        if (s >= win) {
            sum = __fsub_rn(sum, ring[lane][(s - win) & (kRing - 1)]);
            if (s >= len) --cur;
        }
        if (s >= half - 1) {
            const int oi = s - (half - 1);
            bool keep = true;
            int slot = oi;
            if (nsel) {
                keep = (next_j < nsel) && (oi == next_sel);
                slot = next_j;
            }
            if (keep) {
                float o;
                if ((cur & (cur - 1)) == 0)
                    o = __fmul_rn(sum, 1.0f / (float)cur);  // exact: power-of-two divisor
                else
                    o = __fdiv_rn(sum, (float)cur);
                if (my_line < lines) dst[(long long)slot * out_lines + my_line] = o;
                if (nsel) {
                    ++next_j;
                    next_sel = (int)(((next_j + 0.5) * len) / 64);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------
// Fused down-sampler for the reference's frame geometry, 512x512 (window 4 on both axes):
// luma + 2 x (rows, cols) Jarosz passes + decimation in ONE kernel, the frame read from HBM
// exactly once and every intermediate kept in LDS.
//
// One workgroup of 512 lanes per frame walks the frame in 17 strips of 32 columns:
//   A  (all lanes, lane = row)      rep-1 filter along the row, 32 steps, state in registers;
//                                   outputs land in buf[row][c], c <-> column 32k-2+c (the
//                                   filter's output lags its input by 2)
//   B  (wave 0, lane = column)      rep-1 filter down each buffer column, IN PLACE (the lagging
//                                   operand comes from a 4-register ring, so a row is never read
//                                   after it was overwritten)
//   C  (all lanes, lane = row)      rep-2 filter along the row over the buffer columns; emits only
//                                   the 4 columns per strip that the decimation samples
//   D  (wave 1, lanes 0..3)         rep-2 filter down each sampled column; emits only the 64
//                                   sampled rows -> out64[frame][i][j]; runs concurrently with
//                                   B of the next strip
// Every filter is upstream's sequential running sum (box1DFloat) with identical operation
// order, so the result is bit-identical to the 4-launch generic path and to the oracle.
constexpr int kF = 512;        // frame side
constexpr int kS = 32;         // strip width
constexpr int kCsLd = kF + 1;

template <int CH, int S = kS>
struct StripRaw {
    uint4 q[S * CH / 16];
};

template <int CH, int S = kS>
__device__ __forceinline__ void load_strip_raw(const uint8_t* __restrict__ row_ptr, int k, StripRaw<CH, S>& raw) {
    const uint4* p = reinterpret_cast<const uint4*>(row_ptr + (S * CH) * k);  // S px, 16-B aligned
#pragma unroll
    for (int q = 0; q < S * CH / 16; ++q) raw.q[q] = p[q];
}

template <int CH, int S = kS>
__device__ __forceinline__ void strip_luma(const StripRaw<CH, S>& raw, float (&v)[S]) {
    if (CH == 3) {
        uint32_t w[S * 3 / 4];
#pragma unroll
        for (int q = 0; q < S * 3 / 16; ++q) {
            const uint4 t = raw.q[q];
            w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int c = 0; c < S; ++c) {
            const int b0 = 3 * c, b1 = 3 * c + 1, b2 = 3 * c + 2;
            const float r = (float)((w[b0 >> 2] >> (8 * (b0 & 3))) & 0xFFu);
            const float g = (float)((w[b1 >> 2] >> (8 * (b1 & 3))) & 0xFFu);
            const float b = (float)((w[b2 >> 2] >> (8 * (b2 & 3))) & 0xFFu);
            float yv = __fmul_rn(0.299f, r);
            yv = __fadd_rn(yv, __fmul_rn(0.587f, g));
            yv = __fadd_rn(yv, __fmul_rn(0.114f, b));
            v[c] = yv;
        }
    } else {
        uint32_t w[S / 4];
#pragma unroll
        for (int q = 0; q < S / 16; ++q) {
            const uint4 t = raw.q[q];
            w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int c = 0; c < S; ++c) v[c] = luma_gray((w[c >> 2] >> (8 * (c & 3))) & 0xFFu);
    }
}

// One 1-D Jarosz pass (window 4, 512 elements) down an LDS column, as a single lane's
// sequential recurrence. INPLACE (pass B): every output is written back to the column, two
// rows behind the read position (the lagging operand lives in registers, so a row is never
// read after it was overwritten). Otherwise (pass D): only the decimation samples
// oy = 8i+4 (produced at step s = 8i+6, full window) go to dst[i*64 + j].
// The main loop handles 8 steps per trip with the next 8 values already in flight; the code is
// branch-free inside a trip because a lone wave is issue-bound here.
#define HVD_COL_STEP(X, L)            \
    sum = __fadd_rn(sum, (X));        \
    sum = __fsub_rn(sum, (L));        \
    (L) = (X);

template <bool INPLACE>
__device__ __forceinline__ void column_pass512(float* col, const int stride, float* __restrict__ dst, const int j) {
    float x0 = col[0], x1 = col[stride], x2 = col[2 * stride], x3 = col[3 * stride];
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = col[(4 + e) * stride];
    float sum = __fadd_rn(__fadd_rn(x0, x1), x2);  // ((0 + x0) + x1) + x2; 0 + x0 is exact
    if (INPLACE) col[0] = __fdiv_rn(sum, 3.0f);
    sum = __fadd_rn(sum, x3);
    if (INPLACE) col[stride] = __fmul_rn(sum, 0.25f);
    float l0 = x0, l1 = x1, l2 = x2, l3 = x3;

    // steps s0 .. s0+7 with the values in v[]; outputs are rows s0-2 .. s0+5
#define HVD_COL_CHUNK(V, S0)                                                                   \
    {                                                                                          \
        float o[8];                                                                            \
        HVD_COL_STEP(V[0], l0) o[0] = sum; HVD_COL_STEP(V[1], l1) o[1] = sum;                  \
        HVD_COL_STEP(V[2], l2) o[2] = sum; HVD_COL_STEP(V[3], l3) o[3] = sum;                  \
        HVD_COL_STEP(V[4], l0) o[4] = sum; HVD_COL_STEP(V[5], l1) o[5] = sum;                  \
        HVD_COL_STEP(V[6], l2) o[6] = sum; HVD_COL_STEP(V[7], l3) o[7] = sum;                  \
        if (INPLACE) {                                                                         \
            _Pragma("unroll") for (int e = 0; e < 8; ++e)                                      \
                col[((S0) - 2 + e) * stride] = __fmul_rn(o[e], 0.25f);                         \
        } else {                                                                               \
            dst[(((S0) - 4) >> 3) * 64 + j] = __fmul_rn(o[2], 0.25f); /* step S0+2 = 8i+6 */   \
        }                                                                                      \
    }

    // main: s0 = 4, 12, ..., 492 (62 trips of 8 steps, unrolled in pairs so that a/b swap statically)
#pragma unroll 1
    for (int s0 = 4; s0 < 500; s0 += 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = col[(s0 + 8 + e) * stride];
        HVD_COL_CHUNK(a, s0)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = col[(s0 + 16 + e) * stride];
        HVD_COL_CHUNK(b, s0 + 8)
    }
    // here a[] holds rows 500..507; rows 508..511 remain
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = col[(508 + e) * stride];
    HVD_COL_CHUNK(a, 500)
    {
        float o[4];
        HVD_COL_STEP(t[0], l0) o[0] = sum; HVD_COL_STEP(t[1], l1) o[1] = sum;
        HVD_COL_STEP(t[2], l2) o[2] = sum; HVD_COL_STEP(t[3], l3) o[3] = sum;
        if (INPLACE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) col[(506 + e) * stride] = __fmul_rn(o[e], 0.25f);
        } else {
            dst[63 * 64 + j] = __fmul_rn(o[2], 0.25f);  // step 510 = 8*63 + 6
        }
    }
    if (INPLACE) {  // box1DFloat phase 4: outputs 510 (/3) and 511 (/2)
        sum = __fsub_rn(sum, l0);
        col[510 * stride] = __fdiv_rn(sum, 3.0f);
        sum = __fsub_rn(sum, l1);
        col[511 * stride] = __fmul_rn(sum, 0.5f);
    }
#undef HVD_COL_CHUNK
}
#undef HVD_COL_STEP

// (Measured and dropped, profiles/r01_pmc_down512.txt: pass D as its own kernel -- same speed; 64-column strips with
// one workgroup per CU -- 15 % slower; a 9-wave systolic form with LDS mailboxes -- same speed. The lever turned out
// to be a different decomposition altogether: k_down512w below.)
// S = strip width. 32: 2 workgroups per CU (75.8 KB of LDS, <= 128 VGPRs), the throughput form for batches that fill the chip.
// 64 (round 5): ONE workgroup per CU (149.5 KB), half as many strips -- and pass B, a lone wave walking 512 rows, is what a
// frame's latency is made of, so a batch of up to 256 frames (one workgroup per CU either way: the tail batch of a streamed
// video, a caller with a handful of frames) takes 0.12 instead of 0.19 ms (scripts/gpu_down512_small.py). At full load the wide
// form is 15 % slower (round 1), hence the batch-size rule in launch_pdq_downsample.
template <int CH, int S>
__global__ __launch_bounds__(512, S == 32 ? 4 : 2) void k_down512(
    const uint8_t* __restrict__ frames, long long n, float* __restrict__ out64) {
    constexpr int NST = kF / S;       // full strips; strip NST holds the two tail columns
    constexpr int SPS = S / 8;        // decimation samples per strip
    __shared__ float buf[kF][S + 1];
    __shared__ float cs[SPS][kCsLd];
    const int y = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;

    for (long long f = blockIdx.x; f < n; f += gridDim.x) {
        const uint8_t* row_ptr = frames + (size_t)f * kF * kF * CH + (size_t)y * kF * CH;
        float* dst = out64 + (size_t)f * 4096;
        float sA = 0.0f, lagA[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float sC = 0.0f, lagC[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        StripRaw<CH, S> raw;
        load_strip_raw<CH, S>(row_ptr, 0, raw);

#pragma unroll 1
        for (int k = 0; k <= NST; ++k) {
            // ---------------- A: rep-1 along the row -------------------------------------
            if (k < NST) {
                float v[S];
                strip_luma<CH, S>(raw, v);
                if (k < NST - 1) load_strip_raw<CH, S>(row_ptr, k + 1, raw);  // in flight during B/C of this strip
                if (k == 0) {
                    // s = 0,1: accumulate only; s = 2: /3; s = 3: /4 (box1DFloat phases 1-2)
                    sA = __fadd_rn(sA, v[0]);
                    sA = __fadd_rn(sA, v[1]);
                    sA = __fadd_rn(sA, v[2]);
                    buf[y][2] = __fdiv_rn(sA, 3.0f);
                    sA = __fadd_rn(sA, v[3]);
                    buf[y][3] = __fmul_rn(sA, 0.25f);
#pragma unroll
                    for (int c = 0; c < 4; ++c) lagA[c] = v[c];
#pragma unroll
                    for (int c = 4; c < S; ++c) {
                        sA = __fadd_rn(sA, v[c]);
                        sA = __fsub_rn(sA, lagA[c & 3]);
                        lagA[c & 3] = v[c];
                        buf[y][c] = __fmul_rn(sA, 0.25f);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < S; ++c) {
                        sA = __fadd_rn(sA, v[c]);
                        sA = __fsub_rn(sA, lagA[c & 3]);
                        lagA[c & 3] = v[c];
                        buf[y][c] = __fmul_rn(sA, 0.25f);
                    }
                }
            } else {
                // s = 512, 513 (phase 4): outputs 510 (/3) and 511 (/2)
                sA = __fsub_rn(sA, lagA[0]);
                buf[y][0] = __fdiv_rn(sA, 3.0f);
                sA = __fsub_rn(sA, lagA[1]);
                buf[y][1] = __fmul_rn(sA, 0.5f);
            }
            __syncthreads();

            // ---------------- B: rep-1 down the buffer columns (in place) ‖ D of strip k-1 ----
            const int c_lo = (k == 0) ? 2 : 0, c_hi = (k == NST) ? 2 : S;
            if (wave == 0) {
                const int c = lane;
                if (c >= c_lo && c < c_hi) column_pass512<true>(&buf[0][c], S + 1, nullptr, 0);
            } else if (wave == 1 && k > 0) {
                // D: samples written by C of strip k-1: slot jj <-> sample column j = 4(k-1) - 1 + jj
                const int jj = lane, j = SPS * (k - 1) - 1 + jj;
                if (jj < SPS && j >= 0 && j < 64) column_pass512<false>(cs[jj], 1, dst, j);
            }
            __syncthreads();

            // ---------------- C: rep-2 along the row over the buffer columns ------------------
            // buffer column c <-> filter input index t = S*k - 2 + c; ring slot t & 3 = (c + 2) & 3;
            // the output t - 2 is a decimation sample iff c is a multiple of 8 (slot c / 8)
            if (k == 0) {
                // t = 0..S-3 <-> c = 2..S-1
                const float t0 = buf[y][2], t1 = buf[y][3], t2 = buf[y][4], t3 = buf[y][5];
                sC = __fadd_rn(__fadd_rn(__fadd_rn(t0, t1), t2), t3);
                lagC[0] = t0; lagC[1] = t1; lagC[2] = t2; lagC[3] = t3;
#pragma unroll
                for (int c = 6; c < S; ++c) {
                    const float x = buf[y][c];
                    sC = __fsub_rn(__fadd_rn(sC, x), lagC[(c + 2) & 3]);
                    lagC[(c + 2) & 3] = x;
                    if ((c & 7) == 0) cs[c >> 3][y] = __fmul_rn(sC, 0.25f);
                }
            } else if (k < NST) {
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    const float x = buf[y][c];
                    sC = __fsub_rn(__fadd_rn(sC, x), lagC[(c + 2) & 3]);
                    lagC[(c + 2) & 3] = x;
                    if ((c & 7) == 0) cs[c >> 3][y] = __fmul_rn(sC, 0.25f);
                }
            } else {
                // t = 510: output 508 = sample column 63 (slot 0); t = 511 feeds nothing that is sampled
                const float x = buf[y][0];
                sC = __fsub_rn(__fadd_rn(sC, x), lagC[2]);
                cs[0][y] = __fmul_rn(sC, 0.25f);
            }
            __syncthreads();
        }

        // D for the last strip's single sample column (j = 63)
        if (wave == 1 && lane == 0) column_pass512<false>(cs[0], 1, dst, 63);
        __syncthreads();  // cs / buf are reused by the next frame
    }
}



// ---------------------------------------------------------------------------
// k_down512w: the same fused down-sampler with ONE WAVE PER FRAME.
//
// k_down512 spends most of its time with one half-filled wave walking a 512-row strip (pass B)
// while the workgroup's other seven waves wait at a barrier. Here a lone wave owns a frame and
// walks it in tile rows of 64 rows and steps of 32 columns; no wave ever waits for another and the
// only synchronisation is the wave's own LDS ordering. A lone wave issues a dependent VALU
// instruction only every ~8.5 clocks (profiles/r01_ubench_latency.txt), so throughput comes from
// waves per SIMD, i.e. from a small LDS footprint: the transposition buffer is 64 rows x 32 columns
// (8.4 KB => 4 waves per SIMD), and the lanes still all work in every pass because the upper half
// of the wave runs ONE STEP BEHIND the lower half:
//   step tx of tile row ty:  lanes 0..31  <-> rows 64ty+0..31  of column tile tx
//                            lanes 32..63 <-> rows 64ty+32..63 of column tile tx-1
//   A  lane = row     luma of the lane's 32 pixels, rep-1 filter along the row (state in registers
//                     from step to step); output c lands in buf[row][c] <-> column 32t-2+c (lag 2)
//   B  lane = (half, column c)   rep-1 filter down the 32 buffer rows of the lane's half, in place
//                     (buffer row r <-> row 64ty-2+r). The upper half continues the column that the
//                     lower half walked one step earlier (state handed over by a 32-lane shuffle);
//                     across tile rows the state goes through a small per-wave global scratch.
//   C  lane = row     rep-2 filter along the row; only the 4 decimation samples per row and step are
//                     kept (1 KB of LDS)
//   D  lane j = sample column j: rep-2 filter down the rows of its column, 32 rows in the step that
//                     produced them (lower half) and 32 in the next (upper half); the running sum lives in
//                     the lane's registers for the whole frame; emits out64[i][j] at the 8 sampled rows
// Column tile 16 and tile row 8 are the tails of box1DFloat's phase 4 (one column / one row of each is
// needed by the decimation). Starting a line is the steady step on an all-zero state (0 + x and
// x - 0 are exact), so only the two /3 outputs per line are special.
// Scaling: upstream multiplies every window sum by 0.25 before the next pass. Multiplying by a power
// of two is exact and commutes with every later rounding (no overflow/underflow here: values are 0 or
// in [0.114, 255*256]), so the passes hand on the unscaled sums and the product 1/256 is applied once,
// in D; the /3 edge outputs are multiplied by 4 to sit on the same scale. Bit-identical to k_down512,
// the generic path and the oracle.
constexpr int kWR = 64;                                 // rows per tile row (= lanes)
constexpr int kWT = 32;                                 // columns per step
constexpr int kWNX = kF / kWT;                          // full column tiles (16); index 16 = tail
constexpr int kWNY = kF / kWR;                          // full tile rows (8); index 8 = tail
constexpr int kWC = 16;                                 // elements per register chunk of a pass
constexpr int kWStateFloats = (kWNX + 1) * 5 * 32;      // pass-B state: [column tile][5][column]
constexpr int kWScratchFloats = kWStateFloats;

__device__ __forceinline__ void wave_mem_sync() {  // stores of this wave become visible to its other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One chunk of upstream's running sum (window 4): step r adds x[r] and drops the input four steps
// back, which is x[r-4] inside the chunk and lag[r] (the previous chunk's last four inputs) before.
// o[r] is the sum after step r = 4x the filter output two positions behind the input.
template <int N>
__device__ __forceinline__ void w_run(float& sum, float (&lag)[4], const float (&x)[N], float (&o)[N]) {
#pragma unroll
    for (int r = 0; r < N; ++r) {
        sum = __fsub_rn(__fadd_rn(sum, x[r]), r < 4 ? lag[r] : x[r - 4]);
        o[r] = sum;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) lag[k] = x[N - 4 + k];
}

// Global -> register -> LDS staging of the frame bytes. A lane of pass A needs the contiguous bytes of
// ITS row, so a direct load would touch 64 different cache lines per instruction (the TCP then stalls on
// outstanding misses). Instead the wave reads a UNIT = 32 rows (one half of the wave) x two steps
// (64 pixels: 192 bytes rgb, a whole number of 64-byte DRAM sectors, so nothing is fetched twice) as
// consecutive 16-byte pieces (5.3 rows per instruction), parks it in LDS with a padded row stride
// (conflict-free 128-bit reads), and every lane of that half picks up its row. The bytes of the unit's first
// step land in a staging area that aliases the transposition buffer (they are consumed at once); the bytes
// of its second step are parked in 3.5 KB of LDS of their own until the next step. The halves refill on
// alternate steps, so one parking area serves both.
template <int CH>
struct TileLoad {
    static constexpr int SQ = kWT * CH / 16;          // 16-byte pieces per row and step (6 rgb, 2 gray)
    static constexpr int PPR = 2 * SQ;                // pieces per row of a unit
    static constexpr int NI = 32 * PPR / 64;          // load instructions per unit (6 / 2)
    static constexpr int RS = kWT * CH + 16;          // staged row stride in bytes, one step's bytes per row (112 / 48)
};

// Buffer addressing (SGPR resource + 32-bit lane offset + SGPR offset): one VGPR per lane offset instead
// of a 64-bit address pair per access, and out-of-range lanes read 0 / store nothing.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, uint32_t idx) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(idx * 4u), 0, 0));
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, uint32_t idx, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)(idx * 4u), 0, 0);
}


// soff: wave-uniform byte offset of the unit inside the frame. Instruction i = (NI/2) g + sub covers rows
// 16 g .. 16 g + 15 (piece u = 64 sub + lane of that 16-row group), so only NI/2 lane-dependent offsets exist;
// the group offset rides in the scalar offset (loads) or the immediate offset (LDS).
// cache policy of the frame loads
constexpr int kFrameLoadAux = 0;  // cache-policy bits (1 = sc0, 2 = nt, 16 = sc1); nt was measured slower (it also defeats the L2 reuse of shared sectors)
template <int CH>
__device__ __forceinline__ void unit_fetch(__amdgpu_buffer_rsrc_t frame, const uint32_t soff, const int lane,
                                           uint4 (&p)[TileLoad<CH>::NI]) {
    constexpr int PPR = TileLoad<CH>::PPR, SUB = TileLoad<CH>::NI / 2;
    static_assert(16 * PPR == 64 * SUB, "16 rows are a whole number of instructions");
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
        const uint32_t u = 64u * sub + (uint32_t)lane;
        const uint32_t off = (u / PPR) * (uint32_t)(kF * CH) + (u % PPR) * 16u;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(frame, (int)off, (int)(soff + g * 16 * (kF * CH)), kFrameLoadAux);
            p[SUB * g + sub] = make_uint4(v.x, v.y, v.z, v.w);
        }
    }
}

// first-step pieces of every row -> stage, second-step pieces -> park (both LDS, same row stride)
template <int CH>
__device__ __forceinline__ void unit_stage(uint8_t* stage, uint8_t* park, const int lane, const uint4 (&p)[TileLoad<CH>::NI]) {
    constexpr int PPR = TileLoad<CH>::PPR, SUB = TileLoad<CH>::NI / 2, SQ = TileLoad<CH>::SQ;
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
        const uint32_t u = 64u * sub + (uint32_t)lane;
        const uint32_t pp = u % PPR;
        uint8_t* dst = (pp < SQ ? stage : park) + (u / PPR) * TileLoad<CH>::RS + (pp < SQ ? pp : pp - SQ) * 16u;
#pragma unroll
        for (int g = 0; g < 2; ++g) *reinterpret_cast<uint4*>(dst + g * 16 * TileLoad<CH>::RS) = p[SUB * g + sub];
    }
}

template <int N>
__device__ __forceinline__ uint32_t fifo_word(const uint4 (&f)[N], const int k) {  // k is a compile-time constant after unrolling
    const uint4 q = f[k >> 2];
    return (k & 3) == 0 ? q.x : (k & 3) == 1 ? q.y : (k & 3) == 2 ? q.z : q.w;
}
__device__ __forceinline__ float luma_rgb_f(const float r, const float g, const float b) {
    return __fadd_rn(__fadd_rn(__fmul_rn(0.299f, r), __fmul_rn(0.587f, g)), __fmul_rn(0.114f, b));
}

// ds_write_addtid_b32: LDS address = M0 + offset + 4 * lane, no address VGPR. It is the one LDS store that runs at
// 2 cycles per wave-instruction (128 B/clk/CU); ds_write_b32 and ds_write2_b32 cost 4 and 6 because they ship an
// address VGPR per lane over the store path (MI355X_MICROARCH.md, LDS). Round 3 found this kernel bound by exactly that
// path: 64 dwords written per lane and step as ds_write2_b32 = 192 of ~440 LDS cycles per wave-step, 12 waves per CU
// sharing one LDS -> 5300 LDS cycles per CU-step against 3900 VALU cycles per SIMD-step (the "instruction floor" of the
// round-2 ablation was the LDS pipe). The store writes lane-contiguous dwords, i.e. it TRANSPOSES for free: pass A
// (lane = row) writes column-major, pass B (lane = column) reads its column as 128-bit words and writes row-major,
// pass C (lane = row) reads its row as 128-bit words.
// hipcc has no builtin for it; M0 is a reserved register that the compiler re-loads in front of each of its own uses,
// so the asm sets it (one SALU op + the mandatory wait state per group of four stores).
template <int OFF0, int STRIDE>
__device__ __forceinline__ void lds_store4_addtid(uint32_t base, float a, float b, float c, float d) {
    asm volatile(
        "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
        "ds_write_addtid_b32 %0 offset:%5\n\t"
        "ds_write_addtid_b32 %1 offset:%6\n\t"
        "ds_write_addtid_b32 %2 offset:%7\n\t"
        "ds_write_addtid_b32 %3 offset:%8"
        :
        : "v"(a), "v"(b), "v"(c), "v"(d), "s"(base), "n"(OFF0), "n"(OFF0 + STRIDE), "n"(OFF0 + 2 * STRIDE), "n"(OFF0 + 3 * STRIDE)
        : "memory");  // (no "m0" clobber: hipcc 7.2 rejects it as a reserved register -- "may lead to undefined behaviour";
                      // M0 has no value the compiler relies on across statements here: it sets M0 itself, right in front of each
                      // of its own uses -- the global_load_lds / LDS-DMA builtins -- and this kernel has none)
}
template <int OFF0>
__device__ __forceinline__ void lds_store1_addtid(uint32_t base, float a) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tds_write_addtid_b32 %0 offset:%2" : : "v"(a), "s"(base), "n"(OFF0) : "memory");
}

template <int CH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CH == 3 ? 3 : 4))) void k_down512w(
    const uint8_t* __restrict__ frames, long long n, float* __restrict__ out64, float* __restrict__ wscratch) {
    constexpr int NCH = kWT / kWC;      // register chunks per pass (2)
    constexpr int QPC = kWC * CH / 16;  // 16-byte pieces per chunk of a row
    // The transposition buffer holds one step's 64 rows x 32 columns in one of two layouts, both 32 x PL dwords:
    //   column-major (written by A, read by B):  element (row, c)        at c * PL + row
    //   row-major    (written by B, read by C):  element (row = 32h+r, c) at r * PL + 32 h + c
    // PL = 68: 16-byte aligned lines, and the 16 lanes that one ds_read_b128 cycle serves (4-dword accesses at a lane
    // stride of 68 dwords) fall on 16 different groups of 4 banks.
    constexpr int PL = 68;
    __shared__ __attribute__((aligned(16))) float buf[kWT * PL];
    static_assert(sizeof(buf) >= 32 * TileLoad<CH>::RS, "the byte staging aliases the transposition buffer");
    __shared__ __attribute__((aligned(16))) uint8_t park[32 * TileLoad<CH>::RS];
    // this step's C samples [slot % 4][row] alias the transposition buffer: C writes them after its last read of buf.
    // Row stride SMPS = 68 dwords, not 64 (round 6): the four lanes that own a step's four sample columns read the same row
    // index of four different slots in one instruction -- at a stride of 64 dwords those are four addresses in the SAME banks
    // (a 4-way conflict on every one of pass D's reads: the 17 % conflict cycles of profiles/r04_pmc_down512w.txt); 68 moves each
    // slot on by four banks, which is what a 128-bit read per lane needs.
#ifndef HVD_F1_SMPS
#define HVD_F1_SMPS 68
#endif
    constexpr int SMPS = HVD_F1_SMPS;
    float (*smp)[SMPS] = reinterpret_cast<float (*)[SMPS]>(&buf[0]);
    static_assert(sizeof(buf) >= 4 * SMPS * sizeof(float) && SMPS >= kWR && SMPS % 4 == 0, "smp aliases buf");
    const uint32_t buf_lds = (uint32_t)(uintptr_t)(&buf[0]);  // LDS byte address of the buffer (M0 base of the addtid stores)
    constexpr int SQ = TileLoad<CH>::SQ;
    static_assert(SQ == NCH * QPC, "a step is NCH chunks of QPC pieces");
    uint8_t* stage = reinterpret_cast<uint8_t*>(&buf[0]);
    const int lane = threadIdx.x;
    const int half = lane >> 5, cl5 = lane & 31;
    // per-wave scratch: pass-B state [17][5][32]
    const __amdgpu_buffer_rsrc_t rs =
        make_rsrc(wscratch + (size_t)blockIdx.x * kWScratchFloats, kWScratchFloats * sizeof(float));
    constexpr uint32_t row_bytes = kF * CH, frame_bytes = kF * kF * CH;

    // two prefetch buffers: pre[0] always holds the lower half's next unit, pre[1] the upper half's; each is
    // refilled right after it was staged, i.e. TWO steps before it is needed (HBM latency under load exceeds
    // one step; profiles/r01_down512w_ablation.txt)
    uint4 pre[2][TileLoad<CH>::NI];
#pragma unroll
    for (int i = 0; i < TileLoad<CH>::NI; ++i) pre[0][i] = pre[1][i] = make_uint4(0, 0, 0, 0);
    if ((long long)blockIdx.x < n) {
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(frames + (size_t)blockIdx.x * frame_bytes, frame_bytes);
        unit_fetch<CH>(r0, 0, lane, pre[0]);
        unit_fetch<CH>(r0, 32 * row_bytes, lane, pre[1]);
    }
    // (uniform memory schedule, below: in the steady state 34 memory instructions are younger than a unit when it is staged;
    // behind the two units of the prologue there would be 11 and 20, and the compiler's wait in front of EVERY staging would be
    // sized for that path. 32 stores that go nowhere -- past the end of the scratch -- make the prologue the longer path.)
#pragma unroll
    for (int i = 0; i < 32; ++i) buf_st(rs, (uint32_t)(kWScratchFloats + 64 * i), 0.0f);  // (distinct targets: equal stores would be merged)
    uint4 fifo[SQ];  // the bytes of this lane's row for this step
#pragma unroll
    for (int q = 0; q < SQ; ++q) fifo[q] = make_uint4(0, 0, 0, 0);

    // (frames strided by the grid; handing them out dynamically, as k_pdq_hash64 does for long launches, was measured:
    // no gain at 6144 frames -- the kernel waits on the memory side -- and a longer tail at sizes that are not a multiple
    // of the resident waves)
    // UNIFORM MEMORY SCHEDULE (round 6). vmcnt retires in order and hipcc's s_waitcnt bookkeeping takes, for every wait, the
    // MINIMUM number of younger memory instructions over all paths that reach it. Until round 5 some steps issued the frame
    // fetch and some did not, stores sat under lane conditions, and the first state load of a tile row stood in front of the
    // step loop -- so the minimum was 0 and every step began with s_waitcnt vmcnt(0): the frame fetch issued one step earlier
    // (wanted two steps later) and the acknowledgements of the step's stores were waited for on the spot (ablations in
    // profiles/r06_down512w_ablation.txt: no state loads -3.6 %, no state traffic -11 %, no D stores -11 %). Now EVERY step issues
    // the same sequence -- 5 state loads, the fetch slot's loads (one unit into pre[step parity]: a zero-byte resource where
    // there is nothing to fetch), 5 state stores, 4 output stores (lanes with nothing to store aim past the end of their
    // resource) -- so the compiler's counts are exact and a wait leaves everything younger than what it needs in flight.
    float nxB = 0.0f, nxl[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the lower half's pass-B state for the NEXT step, in flight
    for (long long f = blockIdx.x; f < n; f += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rd = make_rsrc(out64 + (size_t)f * 4096, 4096 * sizeof(float));
        float sD = 0.0f, dl[4] = {0.0f, 0.0f, 0.0f, 0.0f};

#pragma unroll 1
        for (int ty = 0; ty <= kWNY; ++ty) {
            float sA = 0.0f, al[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            float sC = 0.0f, cl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            // pass-B state a lane starts its step with: the upper half gets it from the lower half's previous step (a
            // 32-lane shuffle at the end of B), the lower half from the tile row above (loaded at the top of the step)
            float inB = 0.0f, inl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            // (the lower half's state for the NEXT step, nxB / nxl, is in flight from the previous step -- for column tile 0
            // from the last step of the tile row above: see the load below; a frame starts from zeros)
            const uint32_t rbase = (uint32_t)(kWR * ty) * row_bytes;  // byte offset of the tile row in the frame

#pragma unroll 1
            for (int tp = 0; tp <= kWNX + 1; tp += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {  // even / odd step: static after unrolling (prefetch buffer, refilling half)
                const int tx = tp + par;
                const int txl = tx - half;  // the column tile this lane works on in this step
                // pass-B state of the lower half for this step: left by the upper half one tile row up; the
                // load is issued here so that its latency hides behind pass A
                // (vmcnt retires in order: a wait for these small loads also waits for every load issued before
                // them, so they are issued one step AHEAD of their use and BEFORE this step's frame fetch -- otherwise
                // they would cut the fetch's two steps of flight time down to a fraction of one)
                {
                    if (half == 0) {  // the only read of last step's loads: the wait for them sits here, a step after their issue
                        inB = nxB;
                        inl[0] = nxl[0]; inl[1] = nxl[1]; inl[2] = nxl[2]; inl[3] = nxl[3];
                    }
                    // unconditional, all lanes (the upper half's copies are never read): see the tile row's first load
                    // next step = column tile tx + 1 of this tile row (its state was left by the tile row above), or -- from the
                    // tile row's last step -- column tile 0 of the next tile row (left by this one, 16 steps ago)
#ifdef HVD_ABL_NOSTATE
                    const bool have = false;
#else
                    const bool have = tx <= kWNX ? (ty > 0 && tx + 1 <= kWNX) : (ty < kWNY);
#endif
                    const uint32_t pi = (have ? (uint32_t)(tx <= kWNX ? tx + 1 : 0) * (5 * 32) : (uint32_t)kWScratchFloats) + (uint32_t)cl5;
#ifdef HVD_ABL_NOSTATELOAD  // timing ablation only (wrong results): no state load at all -> no vmcnt wait at the top of a step
                    (void)pi;
                    nxB = 0.0f; nxl[0] = nxl[1] = nxl[2] = nxl[3] = 0.0f;
#else
                    nxB = buf_ld(rs, pi);
                    nxl[0] = buf_ld(rs, pi + 32); nxl[1] = buf_ld(rs, pi + 64);
                    nxl[2] = buf_ld(rs, pi + 96); nxl[3] = buf_ld(rs, pi + 128);
#endif
                }
                // ---------------- A: luma + rep-1 along the row (lane = row 64ty + lane) -----------------
                // unit m of half `par` arrives in this step (lower half: steps 2m, 2m+1 <-> tiles 2m, 2m+1; upper half: steps
                // 2m+1, 2m+2); the other half moves on to the second step of its unit
                const int hr = par;
                const bool stage_now = ty < kWNY && tx < kWNX;
                if (ty < kWNY && tx <= kWNX) {
                    if (half != hr) {  // second step of this half's unit: parked one step ago
#pragma unroll
                        for (int q = 0; q < SQ; ++q)
                            fifo[q] = *reinterpret_cast<const uint4*>(park + cl5 * TileLoad<CH>::RS + q * 16);
                    }
                    wave_mem_sync();  // ... before the arriving unit is parked there
                    if (stage_now) unit_stage<CH>(stage, park, lane, pre[par]);
                }
                {   // THE FETCH SLOT: every step issues the six loads of one unit into pre[par] (uniform memory schedule, above) --
                    // ONE load site whose frame, offset and SIZE are scalar choices (alternative load sites merged into pre[par]
                    // made the compiler load into temporaries and copy them over at the join, behind an s_waitcnt vmcnt(0)).
                    //   steps 0 .. 13 of a tile row: this half's next unit, staged two steps from now
                    //   steps 14, 15: nothing (a zero-byte resource answers zeros and fetches nothing; pre[par] is dead here)
                    //   steps 16, 17: this half's first unit of the next tile row -- from the tail tile row: of the next frame --
                    //                 again two steps ahead of its use (until round 5 it was fetched at steps 14, 15, and the two
                    //                 tail steps issued nothing: their waits then counted no fetch in flight, see above)
                    long long fsel = f;
                    uint32_t fbytes = 0, soff = 0;
                    if (ty < kWNY) {
                        if (tx + 2 < kWNX) {
                            fbytes = frame_bytes;
                            soff = rbase + (uint32_t)(32 * par) * row_bytes + (uint32_t)((tx + 2 - par) >> 1) * (2 * kWT * CH);
                        } else if (tx >= kWNX && ty + 1 < kWNY) {
                            fbytes = frame_bytes;
                            soff = rbase + (uint32_t)(kWR + 32 * par) * row_bytes;
                        }
                    } else if (tx >= kWNX && f + gridDim.x < n) {
                        fsel = f + gridDim.x;
                        fbytes = frame_bytes;
                        soff = (uint32_t)(32 * par) * row_bytes;
                    }
#ifdef HVD_ABL_NOFETCH  // timing ablation only: every prefetch reads from a zero-byte resource
                    fbytes = 0;
#endif
                    unit_fetch<CH>(make_rsrc(frames + (size_t)fsel * frame_bytes, fbytes), soff, lane, pre[par]);
                }
                if (ty < kWNY) {
                    float tailA = 0.0f;
                    if (tx >= kWNX) {  // phase 4: output 510 = (sum - x[508]) / 3; output 511 is never sampled
                        tailA = __fmul_rn(__fdiv_rn(__fsub_rn(sA, al[0]), 3.0f), 4.0f);
                    }
                    if (tx <= kWNX) {
                        if (stage_now) {
                            wave_mem_sync();
                            if (half == hr) {
#pragma unroll
                                for (int q = 0; q < SQ; ++q)
                                    fifo[q] = *reinterpret_cast<const uint4*>(stage + cl5 * TileLoad<CH>::RS + q * 16);
                            }
                            wave_mem_sync();  // the staged bytes are in registers before A overwrites them
                        }
                        if (tx <= 1 && txl == 0) {  // a new line starts from the all-zero state
                            sA = 0.0f;
                            al[0] = al[1] = al[2] = al[3] = 0.0f;
                        }
                        // 4 pixels at a time (12 bytes rgb / 4 bytes gray), straight from the FIFO words: few live
                        // registers, and the scheduler may overlap the luma of one group with the running sum of the
                        // previous one (the fence every two groups keeps it from unrolling the whole step into registers)
#pragma unroll
                        for (int j = 0; j < kWT / 4; ++j) {
                            float v[4];
#ifdef HVD_ABL_NOLUMA  // timing ablation only (wrong results): one conversion per pixel instead of the 8-instruction luma
                            if (CH == 3) {
                                const uint32_t w0 = fifo_word(fifo, 3 * j), w1 = fifo_word(fifo, 3 * j + 1), w2 = fifo_word(fifo, 3 * j + 2);
                                v[0] = (float)(w0 & 0xFFu); v[1] = (float)(w0 >> 24); v[2] = (float)((w1 >> 16) & 0xFFu); v[3] = (float)((w2 >> 8) & 0xFFu);
                            } else
#endif
                            if (CH == 3) {
                                const uint32_t w0 = fifo_word(fifo, 3 * j), w1 = fifo_word(fifo, 3 * j + 1), w2 = fifo_word(fifo, 3 * j + 2);
                                v[0] = luma_rgb_f((float)(w0 & 0xFFu), (float)((w0 >> 8) & 0xFFu), (float)((w0 >> 16) & 0xFFu));
                                v[1] = luma_rgb_f((float)(w0 >> 24), (float)(w1 & 0xFFu), (float)((w1 >> 8) & 0xFFu));
                                v[2] = luma_rgb_f((float)((w1 >> 16) & 0xFFu), (float)(w1 >> 24), (float)(w2 & 0xFFu));
                                v[3] = luma_rgb_f((float)((w2 >> 8) & 0xFFu), (float)((w2 >> 16) & 0xFFu), (float)(w2 >> 24));
                            } else {
                                const uint32_t w0 = fifo_word(fifo, j);
#pragma unroll
                                for (int k = 0; k < 4; ++k) v[k] = luma_gray((w0 >> (8 * k)) & 0xFFu);
                            }
                            float o[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                sA = __fsub_rn(__fadd_rn(sA, v[k]), al[k]);
                                o[k] = sA;
                                al[k] = v[k];
                            }
                            if (j == 0 && tx <= 1) {  // output 0 of the line = (x0 + x1 + x2) / 3
                                if (txl == 0) o[2] = __fmul_rn(__fdiv_rn(o[2], 3.0f), 4.0f);
                            }
                            // column-major: element (row = lane, column 4j + k) at (4j + k) * PL + lane
                            switch (j) {  // (j is a constant after unrolling; the offsets are instruction immediates)
                                case 0: lds_store4_addtid<0 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 1: lds_store4_addtid<1 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 2: lds_store4_addtid<2 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 3: lds_store4_addtid<3 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 4: lds_store4_addtid<4 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 5: lds_store4_addtid<5 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 6: lds_store4_addtid<6 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                default: lds_store4_addtid<7 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                            }
                            if (j & 1) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (tx >= kWNX && txl == kWNX) lds_store1_addtid<0>(buf_lds, tailA);  // element (row = lane, column 0)
                }
                wave_mem_sync();

                // ---------------- B: rep-1 down the buffer columns, in place (lane = half, column) --------
                {
                    const bool valid = (txl == 0) ? (cl5 >= 2) : (txl == kWNX) ? (cl5 == 0) : (txl > 0 && txl < kWNX);
                    const uint32_t sti = (uint32_t)(txl < 0 ? 0 : txl) * (5 * 32) + (uint32_t)cl5;  // index into the state scratch
                    float stS = 0.0f, stl[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // what the upper half leaves for the tile row below
                    bool st_ok = false;
                    if (ty < kWNY) {
                        float sB = inB, bl[4] = {inl[0], inl[1], inl[2], inl[3]};
                        // my column (cl5), my half's 32 rows: one contiguous run of the column-major buffer
                        float x[32];
                        {
                            const float4* cp = reinterpret_cast<const float4*>(&buf[cl5 * PL + 32 * half]);
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const float4 v = cp[q];
                                x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
                            }
                        }
                        wave_mem_sync();  // every lane holds its inputs: the buffer may change layout under them
#pragma unroll
                        for (int g = 0; g < 8; ++g) {
                            float o[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int r = 4 * g + k;
                                sB = __fsub_rn(__fadd_rn(sB, x[r]), r < 4 ? bl[r] : x[r - 4]);
                                o[k] = sB;
                            }
                            if (g == 0 && ty == 0) {  // output row 0 = (x0 + x1 + x2) / 3, lower half only
                                if (half == 0) o[2] = __fmul_rn(__fdiv_rn(o[2], 3.0f), 4.0f);
                            }
                            // row-major: element (row = 32 half + r, column cl5) at r * PL + (32 half + cl5) = r * PL + lane
                            switch (g) {
                                case 0: lds_store4_addtid<0 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 1: lds_store4_addtid<1 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 2: lds_store4_addtid<2 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 3: lds_store4_addtid<3 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 4: lds_store4_addtid<4 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 5: lds_store4_addtid<5 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                case 6: lds_store4_addtid<6 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                                default: lds_store4_addtid<7 * 16 * PL, 4 * PL>(buf_lds, o[0], o[1], o[2], o[3]); break;
                            }
                            if (g & 1) __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) bl[k] = x[28 + k];
                        stS = sB;
#pragma unroll
                        for (int k = 0; k < 4; ++k) stl[k] = bl[k];
#ifndef HVD_ABL_NOSTATE
                        st_ok = half && valid;
#endif
                        // lower half -> upper half of the next step (lane l -> lane l + 32)
                        inB = __shfl_up(sB, 32, 64);
#pragma unroll
                        for (int k = 0; k < 4; ++k) inl[k] = __shfl_up(bl[k], 32, 64);
                    } else if (half == 0 && valid) {  // phase 4, row 510 only (buffer row 0)
                        const float sB = __fsub_rn(inB, inl[0]);
                        lds_store1_addtid<0>(buf_lds, __fmul_rn(__fdiv_rn(sB, 3.0f), 4.0f));  // row-major element (row 0, column cl5 = lane)
                    }
                    {  // every step, every lane (uniform memory schedule): lanes with nothing to leave aim past the scratch's end
                        const uint32_t si = st_ok ? sti : (uint32_t)kWScratchFloats;
                        buf_st(rs, si, stS);
                        buf_st(rs, si + 32, stl[0]); buf_st(rs, si + 64, stl[1]);
                        buf_st(rs, si + 96, stl[2]); buf_st(rs, si + 128, stl[3]);
                    }
                }
                wave_mem_sync();

                // ---------------- C: rep-2 along the row over the buffer columns (lane = buffer row) -------
                // buffer column c <-> input index X = 32t-2+c; output X-2 is decimation sample j = 4t-1+c/8 iff c
                // is a multiple of 8 -> sample slot 4t + c/8 (slot 0 = j -1 does not exist). The step's 4 samples
                // of every row go to smp[slot % 4][row] in LDS for pass D below. Lanes whose row does not exist
                // (ty = 0: r < 2; ty = 8: r > 0) compute on leftovers; D ignores them.
                {
                    float sv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    const bool tail = (tx >= kWNX) && (txl == kWNX);
                    const float* rowp = &buf[cl5 * PL + 32 * half];  // my row (lane = 32 half + cl5) of the row-major buffer
                    if (tail) sv[0] = __fsub_rn(__fadd_rn(sC, rowp[0]), cl[0]);  // X = 510: output 508 = sample column 63 (slot 64)
                    const bool store = (txl >= 0) && (txl < kWNX);
                    if (tx <= kWNX) {
                        const bool first = (tx <= 1) && (txl == 0);
                        if (first) {
                            sC = 0.0f;
                            cl[0] = cl[1] = cl[2] = cl[3] = 0.0f;
                        }
#pragma unroll
                        for (int k = 0; k < NCH; ++k) {
                            float y[kWC], o[kWC];
#pragma unroll
                            for (int q = 0; q < kWC / 4; ++q) {
                                const float4 v = reinterpret_cast<const float4*>(rowp + kWC * k)[q];
                                y[4 * q] = v.x; y[4 * q + 1] = v.y; y[4 * q + 2] = v.z; y[4 * q + 3] = v.w;
                            }
                            if (k == 0 && tx <= 1) {  // the line's inputs start at buffer column 2
                                if (first) y[0] = y[1] = 0.0f;
                            }
                            w_run<kWC>(sC, cl, y, o);
                            if (store) {
                                sv[2 * k] = o[0];
                                sv[2 * k + 1] = o[8];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    wave_mem_sync();  // every lane has read its buffer row; the samples may now overwrite it
                    if (store) {  // smp[k][lane]: lane-contiguous, 256 bytes per slot
                        lds_store4_addtid<0, SMPS * 4>(buf_lds, sv[0], sv[1], sv[2], sv[3]);
                    } else if (tail) {
                        lds_store1_addtid<0>(buf_lds, sv[0]);
                    }
                }
                wave_mem_sync();  // buf is rewritten by the next step; smp is read below

                // ---------------- D: rep-2 down the sample columns (lane j owns sample column j = slot j+1) ----
                // Slot group g = slot/4 is produced by the lower half (rows 0..31) in step g and by the upper half
                // (rows 32..63) in step g+1; its four owner lanes run their 32 rows right away, in both steps. The
                // running sum of a column therefore never leaves its lane's registers, within the tile row and from
                // tile row to tile row, and the samples never leave LDS (a global sample ring cost 0.3 MB of
                // memory-side traffic per frame). Only 8 of 64 lanes work here: ~13 % more instructions, the price
                // of not transposing through memory. Row r <-> input row Y = 64ty-2+r; output Y-2 is decimation row
                // i = 8ty-1+r/8 iff r % 8 == 0.
                {
                    const int js = lane + 1, gs = js >> 2;
#ifdef HVD_ABL_NOD  // timing ablation only (wrong results): pass D does nothing
                    const bool lo_pass = false, hi_pass = false;
#else
                    const bool lo_pass = (gs == tx) && (tx <= kWNX), hi_pass = (gs == tx - 1);
#endif
                    // the step's (up to) four outputs of a lane and where they go; 4096 = past the end of the frame's 64 x 64
                    // floats = nowhere (every step issues its four stores: uniform memory schedule)
                    uint32_t di[4] = {4096u, 4096u, 4096u, 4096u};
                    float dv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (lo_pass || hi_pass) {
                        const int r0 = hi_pass ? 32 : 0;
                        const float* zp = &smp[js & 3][r0];
                        if (ty < kWNY) {
#pragma unroll
                            for (int k = 0; k < 32 / kWC; ++k) {
                                float z[kWC], o[kWC];
#pragma unroll
                                for (int r = 0; r < kWC; ++r) z[r] = zp[kWC * k + r];
                                if (k == 0 && ty == 0) {  // the line's inputs start at row 2 of the first tile row
                                    if (lo_pass) z[0] = z[1] = 0.0f;
                                }
                                w_run<kWC>(sD, dl, z, o);
                                const int ib = 8 * ty - 1 + ((r0 + kWC * k) >> 3);
                                if (ib >= 0) di[2 * k] = (uint32_t)(ib * 64 + lane);
                                dv[2 * k] = __fmul_rn(o[0], 0x1p-8f);
                                di[2 * k + 1] = (uint32_t)((ib + 1) * 64 + lane);
                                dv[2 * k + 1] = __fmul_rn(o[8], 0x1p-8f);
                            }
                        } else if (lo_pass) {  // Y = 510 (row 0 of the tail tile row): output 508 = decimation row 63
                            sD = __fsub_rn(__fadd_rn(sD, zp[0]), dl[0]);
                            di[0] = (uint32_t)(63 * 64 + lane);
                            dv[0] = __fmul_rn(sD, 0x1p-8f);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) buf_st(rd, di[i], dv[i]);
                }
                wave_mem_sync();  // smp is rewritten by the next step
            }
            }
        }
    }
}

}  // namespace

namespace hvd {

int g_pdq_dct_mode = 0;           // 0: strict mul-then-add on the VALU (default); 1: fma chain on the matrix cores
int g_pdq_hash_grid = 0;          // A/B switch (hvd_debug_set "pdq_hash_grid"): workgroups of k_pdq_hash64, 0 = default
int g_pdq_luma_lut = 1;           // 0: compute luma, 1: LDS table, 2: LDS table, loads in groups of 16
// stage-1 DCT operand source (hvd_debug_set "pdq_dct_from_lds"): 0 SGPRs, 1 LDS, 2 literals, 3 (default) by batch size --
// literals from 64k frames on (+9 % at 400k frames: full-rate multiplies), SGPRs below (the unrolled 21 KB of code cost
// 4 % at 10k frames, where every workgroup runs it once or twice; profiles/r02_k1_dct_operand.txt)
int g_pdq_dct_from_lds = 3;

void pdq_dct_table_copy(float* out_16x64) {
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 64; ++k) memcpy(&out_16x64[i * 64 + k], &kDctBits[i][k], 4);
}

bool pdq_dct_table_matches(const float* host_16x64) {
    for (int i = 0; i < 16; ++i)
        for (int k = 0; k < 64; ++k) {
            uint32_t w;
            memcpy(&w, &host_16x64[i * 64 + k], 4);
            if (w != kDctBits[i][k]) return false;
        }
    return true;
}

// Work counters of k_pdq_hash64: a ring of slots, one per launch in flight -- the streaming hasher runs up to three launches
// concurrently on its own streams. A slot is zeroed ON THE LAUNCH'S STREAM right in front of the launch (the
// self-cleaning alone -- the launch's last draw zeroes its slot -- left a non-zero slot behind a faulted or aborted launch,
// and the launch handed that slot 256 launches later skipped chunks silently; the one-off clearing also ran on the null
// stream, unordered with the non-blocking streams the kernels run on). ~2 us per launch of >= 64 k frames.
constexpr int kMaxWorkDevices = 16;  // one ring per HIP device (contexts that share a device share its ring: a slot is per launch)
static std::atomic<unsigned int*> g_hash_work[kMaxWorkDevices] = {};
static std::atomic<unsigned int> g_hash_work_next{0};
constexpr unsigned int kHashWorkSlots = 256;

void pdq_release() {
    for (auto& w : g_hash_work) {
        unsigned int* p = w.exchange(nullptr);
        if (p) (void)hipFree(p);
    }
}

// the next counter slot of the current device's ring (allocated on first use), zeroed in stream order
static hipError_t work_slot(unsigned int** out, hipStream_t s) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxWorkDevices) return hipErrorInvalidDevice;
    unsigned int* base = g_hash_work[dev].load(std::memory_order_acquire);
    if (!base) {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        base = g_hash_work[dev].load(std::memory_order_acquire);
        if (!base) {
            unsigned int* p = nullptr;
            e = hipMalloc((void**)&p, kHashWorkSlots * 2 * sizeof(unsigned int));
            if (e != hipSuccess) return e;
            g_hash_work[dev].store(p, std::memory_order_release);
            base = p;
        }
    }
    *out = base + 2u * (g_hash_work_next.fetch_add(1u) % kHashWorkSlots);
    return hipMemsetAsync(*out, 0, 2 * sizeof(unsigned int), s);
}

hipError_t launch_pdq_hash64(const void* d_in, int kind, int64_t n, const float* d_dct, uint8_t* d_hashes,
                             int32_t* d_quality, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int64_t groups = (n + kWaves - 1) / kWaves;
    // literals from 8 k frames on (round 3: with the leaner quality term the literal form also wins at 10 k frames)
    const int dlds = g_pdq_dct_from_lds == 3 ? (n >= 8192 ? 2 : 0) : g_pdq_dct_from_lds;
    const int lut = g_pdq_luma_lut;
    // long launches of the strict kernels distribute their work dynamically (see the kernel), several groups per draw
    const bool dynamic = g_pdq_dct_mode != 1 && n >= 65536;
    unsigned int* work = nullptr;
    if (dynamic) {
        hipError_t e = work_slot(&work, s);
        if (e != hipSuccess) return e;
    }
    const int chunk = !dynamic ? 1 : n >= (1 << 20) ? 8 : n >= (1 << 18) ? 4 : 1;
    const int64_t nchunks = (groups + chunk - 1) / chunk;
    // 92-100 VGPRs = 4 (5 for the literal form) waves per SIMD. Static stride below 64 k frames: the grid is what is resident
    // at once, 4 workgroups per CU, so that no second dispatch round of a few workgroups trails the launch (10 k frames: 1024
    // workgroups 51-52 us, 1792 54-55 us); dynamic: 7 per CU (the LDS limit; 1792 measured best); fma kernel: static, 7 per CU.
    // (16 k - 64 k frames: static stride over 7 per CU -- more workgroups than are resident, so the hardware's own
    // dispatch evens the load out: 20 k frames 97 -> 86 us)
    // Round 4 (the waves of a workgroup no longer wait for each other): below 16 k frames the best static grid is k workgroups
    // per CU with k such that ~60 % of them make a second trip -- 1500 groups: k = 4, 2048: 5, 2500: 6, 3000: 7
    // (profiles/r04_k1_grid.txt: 10 k frames 51-53 us at k = 4, 47.9 us at k = 6) -- i.e. k = ceil(groups / (256 * 1.67)).
    int64_t per_cu = (groups * 3 + 1279) / 1280;
    per_cu = per_cu < 4 ? 4 : per_cu > 7 ? 7 : per_cu;
    const int64_t max_grid = n < 16384 ? 256 * per_cu : 256 * 7;
    dim3 grid((unsigned)(nchunks < max_grid ? nchunks : max_grid));
    if (g_pdq_hash_grid > 0) grid.x = (unsigned)(nchunks < g_pdq_hash_grid ? nchunks : g_pdq_hash_grid);
    if (g_pdq_dct_mode == 1) {
        if (kind == 0)
            hipLaunchKernelGGL(k_pdq_hash64_fma<0>, grid, dim3(256), 0, s, d_in, (long long)n, d_dct, d_hashes, d_quality);
        else
            hipLaunchKernelGGL(k_pdq_hash64_fma<1>, grid, dim3(256), 0, s, d_in, (long long)n, d_dct, d_hashes, d_quality);
        return hipGetLastError();
    }
#define HVD_K1(KIND, D, L) hipLaunchKernelGGL((k_pdq_hash64<KIND, D, L>), grid, dim3(256), 0, s, d_in, (long long)n, d_dct, d_hashes, d_quality, work, chunk)
#define HVD_K1P(D) hipLaunchKernelGGL((k_pdq_hash64<0, D, 1, true>), grid, dim3(256), 0, s, d_in, (long long)n, d_dct, d_hashes, d_quality, work, chunk)
    if (kind == 0 && !dynamic && g_pdq_hash_prefetch && lut == 1 && dlds != 1) {
        if (dlds == 2) HVD_K1P(2);
        else HVD_K1P(0);
    } else if (kind == 0) {
        if (dlds == 2) HVD_K1(0, 2, 1);
        else if (dlds == 1) HVD_K1(0, 1, 1);
        else if (lut == 0) HVD_K1(0, 0, 0);
        else if (lut == 1) HVD_K1(0, 0, 1);
        else HVD_K1(0, 0, 2);
    } else {
        if (dlds == 2) HVD_K1(1, 2, 0);
        else if (dlds == 1) HVD_K1(1, 1, 0);
        else HVD_K1(1, 0, 0);
    }
#undef HVD_K1
#undef HVD_K1P
    return hipGetLastError();
}

int g_pdq_hash_prefetch = 0;  // A/B switch (hvd_debug_set "pdq_hash_prefetch"): measured SLOWER (115 VGPRs = 4 waves per SIMD instead of 5:
                              // 10 k frames 47.6 -> 50.2 us, profiles/r04_k1_grid.txt), so off; bit-identical either way
static int jarosz_window(int dim) { return (dim + 2 * 64 - 1) / (2 * 64); }

// Workspace (floats per frame) the down-sampler needs besides the 64x64 output.
size_t pdq_downsample_ws_floats(int h, int w) { return 2 * (size_t)h * w + (size_t)64 * h; }

bool g_pdq_fused_down512 = true;  // A/B switch (hvd_debug_set "pdq_fused_down512")
int g_pdq_down512_wave = 1;       // k_down512w (one wave per frame): 0 never, 1 for batches >= 704 frames, 2 always
int g_pdq_down512_wave_grid = 0;  // waves in flight; 0 = what is resident at once (rgb: 3 per SIMD, gray: 4)
int g_pdq_down512_strip = 0;      // k_down512's strip width: 0 = 64 for batches of <= 256 frames, else 32; 32 / 64 forced (hvd_debug_set "pdq_down512_strip")

hipError_t launch_pdq_downsample(const uint8_t* d_frames, int64_t n, int h, int w, int channels, float* d_ws,
                                 float* d_out64, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (h == kF && w == kF && g_pdq_fused_down512) {
        const unsigned grid = (unsigned)(n < 512 ? n : 512);  // 2 workgroups per CU fit by LDS (75.8 KB each)
        // one wave per frame (k_down512w) once there are enough frames to fill the chip with lone waves: a wave
        // needs ~0.3 ms per frame whatever the load, a 512-lane workgroup ~0.2 ms, and the chip holds 3072 (rgb) /
        // 4096 (gray) such waves but only 512 such workgroups; crossover measured at ~700 frames
        // (profiles/r01_down512w_ablation.txt). pdq_down512_wave: 0 never, 1 by batch size (default), 2 always.
        const bool use_wave = g_pdq_down512_wave == 2 || (g_pdq_down512_wave == 1 && n >= 704);
        if (use_wave) {
            // d_ws is sized for min(n, 1024) frames of the generic path (2.2 MB each) >= 10.9 KB per wave here.
            // resident waves: rgb 12 per CU (168 VGPRs, 13 KB of LDS), gray 16 per CU (10 KB of LDS)
            const int64_t resident = g_pdq_down512_wave_grid > 0 ? g_pdq_down512_wave_grid : 256 * (channels == 3 ? 12 : 16);
            const int64_t rounds = (n + resident - 1) / resident;  // equal shares: no half-empty last round
            const unsigned gw = (unsigned)((n + rounds - 1) / rounds);
            if (channels == 3)
                hipLaunchKernelGGL(k_down512w<3>, dim3(gw), dim3(64), 0, s, d_frames, (long long)n, d_out64, d_ws);
            else
                hipLaunchKernelGGL(k_down512w<1>, dim3(gw), dim3(64), 0, s, d_frames, (long long)n, d_out64, d_ws);
        } else {
            // wide strips while every frame has a CU of its own (g_pdq_down512_strip: 0 by batch size, 32 / 64 forced)
            const bool wide = g_pdq_down512_strip == 64 || (g_pdq_down512_strip == 0 && n <= 256);
            const unsigned gridw = (unsigned)(n < 256 ? n : 256);
            if (wide && channels == 3)
                hipLaunchKernelGGL((k_down512<3, 64>), dim3(gridw), dim3(512), 0, s, d_frames, (long long)n, d_out64);
            else if (wide)
                hipLaunchKernelGGL((k_down512<1, 64>), dim3(gridw), dim3(512), 0, s, d_frames, (long long)n, d_out64);
            else if (channels == 3)
                hipLaunchKernelGGL((k_down512<3, kS>), dim3(grid), dim3(512), 0, s, d_frames, (long long)n, d_out64);
            else
                hipLaunchKernelGGL((k_down512<1, kS>), dim3(grid), dim3(512), 0, s, d_frames, (long long)n, d_out64);
        }
        return hipGetLastError();
    }
    if (jarosz_window(h) > kTW || jarosz_window(w) > kTW) return hipErrorInvalidValue;
    const size_t hw = (size_t)h * w;
    const int win_rows = jarosz_window(w);  // window of the filter that runs along a row
    const int win_cols = jarosz_window(h);
    // Process frames in slabs so that the workspace stays bounded (y-grid limit too).
    const int64_t slab = 1024;
    for (int64_t f0 = 0; f0 < n; f0 += slab) {
        const int64_t m = (n - f0) < slab ? (n - f0) : slab;
        const size_t cnt = (size_t)(n < slab ? n : slab);  // frames the workspace is sized for
        float* buf1 = d_ws;                   // [m][w][h]  pass-1 output (transposed)
        float* buf2 = d_ws + cnt * hw;        // [m][h][w]  pass-2 output
        float* buf3 = d_ws + 2 * cnt * hw;    // [m][64][h] pass-3 output
        const uint8_t* src = d_frames + (size_t)f0 * hw * channels;
        dim3 g1((h + 63) / 64, (unsigned)m), g2((w + 63) / 64, (unsigned)m), g4(1, (unsigned)m);
        if (channels == 3)
            hipLaunchKernelGGL(k_box_scan_T<3>, g1, dim3(64), 0, s, (const void*)src, buf1, h, w, win_rows, 0,
                               (long long)hw * 3, (long long)hw);
        else
            hipLaunchKernelGGL(k_box_scan_T<1>, g1, dim3(64), 0, s, (const void*)src, buf1, h, w, win_rows, 0,
                               (long long)hw, (long long)hw);
        hipLaunchKernelGGL(k_box_scan_T<0>, g2, dim3(64), 0, s, (const void*)buf1, buf2, w, h, win_cols, 0,
                           (long long)hw, (long long)hw);
        hipLaunchKernelGGL(k_box_scan_T<0>, g1, dim3(64), 0, s, (const void*)buf2, buf3, h, w, win_rows, 64,
                           (long long)hw, (long long)64 * h);
        hipLaunchKernelGGL(k_box_scan_T<0>, g4, dim3(64), 0, s, (const void*)buf3, d_out64 + (size_t)f0 * 4096, 64, h,
                           win_cols, 64, (long long)64 * h, (long long)4096);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_pdq_luma64_rgb(const uint8_t* d_frames, int64_t n, float* d_out64, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const long long npix = (long long)n * 4096;
    long long blocks = (npix + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_luma64_rgb, dim3((unsigned)blocks), dim3(256), 0, s, d_frames, npix, d_out64);
    return hipGetLastError();
}

}  // namespace hvd
