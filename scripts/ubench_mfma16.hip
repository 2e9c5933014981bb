// ubench_mfma16.hip -- register-only inner loop of an all-pairs first stage built on v_mfma_f32_16x16x128_f8f6f4 (one
// instruction = 128 hash bits x 256 pairs) next to the committed 32x32x64 loop (two instructions = 128 bits x 1024 pairs).
// Same flop per pair; the 16x16 form judges tiles of 256 pairs (a false survivor costs one more 16-cycle instruction
// instead of two 32-cycle ones) at the price of 3 VALU ops per 16 MFMA cycles instead of 9 per 64.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma16.hip -o /tmp/ubench_mfma16 && /tmp/ubench_mfma16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int ITERS = 20000;

__device__ __forceinline__ v16f mfma32(const v4i a, const v4i b, const v16f c) {
    const v8i a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0, 0, 0);
}
__device__ __forceinline__ v4f mfma16(const v4i a, const v4i b, const v4f c) {
    const v8i a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 4, 4, 0, 0, 0, 0);
}
__device__ __forceinline__ int or16(const v16f& c) {
    int m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = __float_as_int(c[i]);
    int a0 = m[0] | m[1] | m[2], a1 = m[3] | m[4] | m[5], a2 = m[6] | m[7] | m[8], a3 = m[9] | m[10] | m[11], a4 = m[12] | m[13] | m[14];
    return (a0 | a1 | a2) | (a3 | a4 | m[15]);
}
__device__ __forceinline__ int or4(const v4f& c) {
    return (__float_as_int(c[0]) | __float_as_int(c[1]) | __float_as_int(c[2])) | __float_as_int(c[3]);
}

// MODE 0: committed loop (8 tiles x 2 x 32x32x64 per 32 candidates). MODE 1: 16 tiles x 1 x 16x16x128 per 16 candidates.
// MODE 2: as 1 with TWO panels of 16 candidates interleaved (two B fragments live).
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_loop(int* out, int seed, float c0) {
    uint32_t marks_all = 0;
    if (MODE == 0) {
        constexpr int TILES = 8;
        v4i a[TILES][2];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) a[t][s] = v4i{0x2a2a2a2a ^ (seed + t), (int)threadIdx.x | 0x22222222, 0x22222222 + s, 0x2a222a22};
        v4i b0 = {0x22222222, 0x2a2a2a2a + seed, 0x2a2a2a2a, 0x22222a2a}, b1 = {0x2a222222 + seed, 0x22222222, 0x2a2a2a2a, 0x22222222};
        v16f cinit;
#pragma unroll
        for (int i = 0; i < 16; ++i) cinit[i] = c0;
        for (int it = 0; it < ITERS; ++it) {
            uint32_t marks = 0;
            v16f cur = mfma32(a[0][1], b1, mfma32(a[0][0], b0, cinit));
#pragma unroll
            for (int t = 1; t < TILES; ++t) {
                const v16f nxt = mfma32(a[t][1], b1, mfma32(a[t][0], b0, cinit));
                marks = __builtin_amdgcn_alignbit(marks, (uint32_t)or16(cur), 31);
                cur = nxt;
            }
            marks = __builtin_amdgcn_alignbit(marks, (uint32_t)or16(cur), 31);
            marks_all |= marks;
            b0.x ^= it;
        }
    } else {
        constexpr int TILES = 16;
        v4i a[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) a[t] = v4i{0x2a2a2a2a ^ (seed + t), (int)threadIdx.x | 0x22222222, 0x22222222 + t, 0x2a222a22};
        v4i b0 = {0x22222222, 0x2a2a2a2a + seed, 0x2a2a2a2a, 0x22222a2a}, b1 = {0x2a222222 + seed, 0x22222222, 0x2a2a2a2a, 0x22222222};
        const v4f cinit = {c0, c0, c0, c0};
        // one iteration = 32 candidates = two panels of 16, i.e. the same number of comparisons as MODE 0's iteration
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const v4i b = half ? b1 : b0;
                uint32_t marks = 0;
                v4f cur = mfma16(a[0], b, cinit);
#pragma unroll
                for (int t = 1; t < TILES; ++t) {
                    const v4f nxt = mfma16(a[t], b, cinit);
                    marks = __builtin_amdgcn_alignbit(marks, (uint32_t)or4(cur), 31);
                    cur = nxt;
                }
                marks = __builtin_amdgcn_alignbit(marks, (uint32_t)or4(cur), 31);
                marks_all |= marks;
            }
            b0.x ^= it;
            if (MODE == 2) b1.y ^= it;
        }
    }
    if (marks_all == 0x12345u) out[threadIdx.x] = (int)marks_all;
}

template <int MODE>
int run(const char* name, int* d_out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grid = 256 * 3;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_loop<MODE>, dim3(grid), dim3(256), 0, 0, d_out, rep, 65.0f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double)grid * 4 * ITERS * 16.0 * 131072.0;  // per wave-iteration: 256 rows x 32 candidates x 128 bits x 2
        if (rep) printf("%-52s %8.3f ms  %.2f PFLOP/s (%.3f of 10 PF)\n", name, ms, flop / (ms * 1e-3) / 1e15, flop / (ms * 1e-3) / 1e16);
    }
    return 0;
}

int main() {
    int* d_out; CHK(hipMalloc(&d_out, 4096));
    for (int round = 0; round < 3; ++round) {
        if (run<0>("32x32x64 x2, 8 tiles, or16+alignbit (committed)", d_out)) return 1;
        if (run<1>("16x16x128 x1, 16 tiles, or4+alignbit", d_out)) return 1;
    }
    return 0;
}
