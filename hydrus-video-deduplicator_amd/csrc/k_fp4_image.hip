// k_fp4_image.hip -- the operands of the matrix-core Hamming kernel (k_hamming_mfma.hip) and their preparation.
//
// k_expand_fp4 writes the "FP4 image" of a hash DB: every bit b as the e2m1 number 1 - 2b (0x2 = +1.0, 0xA = -1.0), 128 B per
// hash = 8 chunks of 16 B (chunk c = bits 32c..32c+31 as 32 nibbles); chunk c of hash n lives in slot c ^ ((n >> 1) & 7)
// (hvd_fp4.h: img_slot), so that a wave's ds_read_b128 of one chunk of 32 consecutive hashes is bank-conflict free.
// k_pack_fp4 is the inverse (callers that hand the library only images), and the bit-order kernels rewrite packed hashes and
// image in an order chosen from the library (round 5: the 128 least entangled bits first, for the 128-bit first stage).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "hvd_fp4.h"
#include "hvd_kernels.h"

namespace {

// One thread per (hash, chunk). Rows >= n (padding up to n_pad) become FP4 zeros.
__global__ __launch_bounds__(256) void k_expand_fp4(const uint32_t* __restrict__ db, uint32_t n, uint32_t n_pad,
                                                    uint4* __restrict__ img) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint64_t)n_pad * 8u) return;
    const uint32_t hash = (uint32_t)(idx >> 3), chunk = (uint32_t)(idx & 7u);
    uint32_t o[4] = {0u, 0u, 0u, 0u};
    if (hash < n) {
        const uint32_t w = db[(size_t)hash * 8u + chunk];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t x = 0x22222222u;  // +1.0 (e2m1 code 2) in every nibble
#pragma unroll
            for (int t = 0; t < 8; ++t) x |= ((w >> (8 * d + t)) & 1u) << (4 * t + 3);  // bit set -> sign -> -1.0
            o[d] = x;
        }
    }
    img[(size_t)hash * 8u + img_slot(hash, chunk)] = make_uint4(o[0], o[1], o[2], o[3]);
}

// The inverse: the packed 32-byte hashes of an FP4 image (one thread per (hash, chunk): 32 sign nibbles -> 32 bits). The
// pair-queue form settles its candidates on packed hashes (16 B per half instead of 64); callers that hand the library
// only images (video search, cross search) get them derived here. Rows >= n are not written.
__global__ __launch_bounds__(256) void k_pack_fp4(const uint4* __restrict__ img, uint32_t n, uint32_t* __restrict__ db) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint64_t)n * 8u) return;
    const uint32_t hash = (uint32_t)(idx >> 3), chunk = (uint32_t)(idx & 7u);
    const uint4 v = img[(size_t)hash * 8u + img_slot(hash, chunk)];
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 8; ++t) w |= ((d[k] >> (4 * t + 3)) & 1u) << (8 * k + t);
    db[(size_t)hash * 8u + chunk] = w;
}

// ---- data-dependent bit order (round 5) ---------------------------------------------------------------------------------
// Hamming distance does not care in which order the bits of the hashes are written, as long as it is the same order for
// everybody. The 128-bit first stage does: it sees the first 128 bits, and on real frame hashes some bits move together (on the
// config-5 frames: DCT rows and columns 7..9), so unrelated frames agree on them far more often than on independent bits. The
// video search therefore measures, on a sample of the library, how much every bit correlates with the others, keeps the 128
// least entangled bits for the first stage (hvd_api.cpp: 128 times, drop the bit with the largest sum of |correlation| with the
// bits still in the set) and rewrites packed hashes and FP4 image in that order: 7e-6 of the unrelated pairs pass instead of
// 7.5e-5 (bits 0..63 + 192..255) or 2.3e-4 (bits 0..127). Results are bit-identical by construction.
//   k_bit_rows: the sample, transposed -- rows[b][w] = bit b of sample hashes 64w .. 64w+63 (one ballot per bit and wave)
//   k_bit_cooc: cooc[i][j] = number of sample hashes with bits i and j both set (diagonal: with bit i set)
//   k_reorder_bits: packed hashes -> packed hashes and FP4 image in the new order: bit k of the output = bit perm[k] of the input
__global__ __launch_bounds__(256) void k_bit_rows(const uint32_t* __restrict__ db, uint32_t stride, uint32_t words,
                                                  unsigned long long* __restrict__ rows) {
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (wave >= words) return;  // (wave-uniform)
    const uint32_t* h = db + (size_t)(wave * 64u + lane) * stride * 8u;
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = h[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll 4
        for (int t = 0; t < 32; ++t) {
            const unsigned long long m = __ballot((w[k] >> t) & 1u);
            if (lane == 0u) rows[(size_t)(32 * k + t) * words + wave] = m;
        }
    }
}

__global__ __launch_bounds__(256) void k_bit_cooc(const unsigned long long* __restrict__ rows, uint32_t words,
                                                  uint32_t* __restrict__ cooc) {
    const uint32_t i = blockIdx.x, j = threadIdx.x;
    const unsigned long long* ri = rows + (size_t)i * words;  // (block-uniform: scalar loads)
    const unsigned long long* rj = rows + (size_t)j * words;
    uint32_t c = 0;
    for (uint32_t w = 0; w < words; ++w) c += (uint32_t)__popcll(ri[w] & rj[w]);
    cooc[i * 256u + j] = c;
}

struct BitOrder {
    uint8_t p[256];
};
// One thread per (hash, chunk) as in k_expand_fp4; a block's 32 hashes are staged in LDS, from where every thread gathers
// its 32 bits. Rows >= n of the image (padding up to n_pad) become FP4 zeros; bits_out has n rows.
__global__ __launch_bounds__(256) void k_reorder_bits(const uint32_t* __restrict__ bits_in, uint32_t n, uint32_t n_pad,
                                                      const BitOrder order, uint32_t* __restrict__ bits_out,
                                                      uint4* __restrict__ img) {
    __shared__ uint32_t src[256];
    __shared__ uint8_t perm[256];
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t hash = (uint32_t)(idx >> 3), chunk = (uint32_t)(idx & 7u);
    perm[threadIdx.x] = order.p[threadIdx.x];
    src[threadIdx.x] = hash < n ? bits_in[(size_t)hash * 8u + chunk] : 0u;
    __syncthreads();
    if (idx >= (uint64_t)n_pad * 8u) return;
    uint32_t o[4] = {0u, 0u, 0u, 0u};
    if (hash < n) {
        const uint32_t* mine = src + (threadIdx.x & ~7u);  // the eight words of this thread's hash
        uint32_t w = 0;
#pragma unroll 8
        for (uint32_t t = 0; t < 32u; ++t) {
            const uint32_t b = perm[32u * chunk + t];
            w |= ((mine[b >> 5] >> (b & 31u)) & 1u) << t;
        }
        bits_out[(size_t)hash * 8u + chunk] = w;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t x = 0x22222222u;
#pragma unroll
            for (int t = 0; t < 8; ++t) x |= ((w >> (8 * d + t)) & 1u) << (4 * t + 3);
            o[d] = x;
        }
    }
    img[(size_t)hash * 8u + img_slot(hash, chunk)] = make_uint4(o[0], o[1], o[2], o[3]);
}

}  // namespace

namespace hvd {

uint32_t fp4_rows_padded(uint32_t n) { return ((n ? n : 1u) + 1023u) / 1024u * 1024u; }

hipError_t launch_expand_fp4(const void* d_db, uint32_t n, void* d_img, hipStream_t s) {
    const uint32_t n_pad = fp4_rows_padded(n);
    const uint64_t threads = (uint64_t)n_pad * 8u;
    hipLaunchKernelGGL(k_expand_fp4, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const uint32_t*)d_db, n,
                       n_pad, (uint4*)d_img);
    return hipGetLastError();
}

hipError_t launch_pack_fp4(const void* d_img, uint32_t n, void* d_db, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint64_t threads = (uint64_t)n * 8u;
    hipLaunchKernelGGL(k_pack_fp4, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const uint4*)d_img, n, (uint32_t*)d_db);
    return hipGetLastError();
}

// sample -> co-occurrence counts (k_bit_rows + k_bit_cooc): d_rows 256 * words u64, d_cooc 256 * 256 u32
hipError_t launch_bit_cooc(const void* d_bits, uint32_t stride, uint32_t words, void* d_rows, void* d_cooc, hipStream_t s) {
    hipLaunchKernelGGL(k_bit_rows, dim3((words + 3u) / 4u), dim3(256), 0, s, (const uint32_t*)d_bits, stride, words, (unsigned long long*)d_rows);
    hipLaunchKernelGGL(k_bit_cooc, dim3(256), dim3(256), 0, s, (const unsigned long long*)d_rows, words, (uint32_t*)d_cooc);
    return hipGetLastError();
}

// packed hashes -> packed hashes + FP4 image with bit k = input bit perm[k] (perm MUST be a permutation of 0..255: the caller checks)
hipError_t launch_reorder_bits(const void* d_bits_in, uint32_t n, const uint8_t perm[256], void* d_bits_out, void* d_img, hipStream_t s) {
    const uint32_t n_pad = fp4_rows_padded(n);
    BitOrder o;
    memcpy(o.p, perm, 256);
    const uint64_t threads = (uint64_t)n_pad * 8u;
    hipLaunchKernelGGL(k_reorder_bits, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const uint32_t*)d_bits_in, n, n_pad, o,
                       (uint32_t*)d_bits_out, (uint4*)d_img);
    return hipGetLastError();
}

}  // namespace hvd
