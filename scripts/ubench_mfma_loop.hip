// ubench_mfma_loop.hip -- how busy can the matrix pipe be kept by the inner loop of the FP4 Hamming kernel?
// Registers only (no LDS, no global): per "panel" 8 tiles x 2 v_mfma_f32_32x32x64_f8f6f4 (FP4, 8 passes) and,
// optionally, the 8-instruction v_max3_i32 tree per tile, in different orders. Reports the MFMA rate against
// the nominal peak (one 8-pass MFMA per 32 clk per SIMD at 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_mfma_loop.hip -o /tmp/ubench_mfma && /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int ITERS = 2000, TILES = 8;

__device__ __forceinline__ v16f mfma(const v4i a, const v4i b, const v16f c) {
    const v8i a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0, 0, 0);
}
__device__ __forceinline__ int max16(const v16f& c) {
    int m[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = __float_as_int(c[i]);
    int a0 = max(max(m[0], m[1]), m[2]), a1 = max(max(m[3], m[4]), m[5]), a2 = max(max(m[6], m[7]), m[8]);
    int a3 = max(max(m[9], m[10]), m[11]), a4 = max(max(m[12], m[13]), m[14]);
    return max(max(max(a0, a1), a2), max(max(a3, a4), m[15]));
}

// MODE 0: MFMAs only (accumulators summed at the end). MODE 1: + max tree of tile t after the MFMAs of tile t+1
// (two accumulator sets, the kernel's form). MODE 2: max tree split in halves between the two MFMAs of the next tile.
template <int MODE, int EXTRA>
__global__ __launch_bounds__(256, 2) void k_loop(int* out, int seed) {
    __shared__ v4i lds[2][128 * 8];
    for (int e = threadIdx.x; e < 2 * 128 * 8; e += 256) (&lds[0][0])[e] = v4i{0x22222222, seed + e, 0x2a2a2a2a, e};
    __syncthreads();
    const int li = threadIdx.x & 31, hh = (threadIdx.x >> 5) & 1;
    v4i a[TILES][2];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) a[t][s] = v4i{seed + t, (int)threadIdx.x, s, 0x22222222};
    v4i b0 = {0x22222222, seed, 0x2a2a2a2a, (int)threadIdx.x}, b1 = {seed, 0x22222222, 0x2a2a2a2a, 1};
    const v16f z = {0};
    int mm = 0x80000000;
    v16f keep = z;
    for (int it = 0; it < ITERS; ++it) {
        if (EXTRA >= 1) {
            const v4i* base = &lds[(it >> 2) & 1][((it & 3) * 32 + li) * 8];
            b0 = base[hh ^ (li & 7)];
            b1 = base[(2 + hh) ^ (li & 7)];
            mm = 0x80000000;
        }
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                v16f c = mfma(a[t][0], b0, z);
                c = mfma(a[t][1], b1, c);
                asm volatile("" :: "v"(c));  // keep the result alive without any VALU work
            }
            b0.x += it;
        } else if (MODE == 3) {  // max tree of tile t-2 between the two MFMAs of tile t (three accumulator sets)
            v16f acc[3];
            acc[0] = mfma(a[0][1], b1, mfma(a[0][0], b0, z));
            acc[1] = mfma(a[1][1], b1, mfma(a[1][0], b0, z));
#pragma unroll
            for (int t = 2; t < TILES; ++t) {
                v16f nxt = mfma(a[t][0], b0, z);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                nxt = mfma(a[t][1], b1, nxt);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                mm = max(mm, max16(acc[(t - 2) % 3]));
                acc[t % 3] = nxt;
            }
            mm = max(mm, max16(acc[(TILES - 2) % 3]));
            mm = max(mm, max16(acc[(TILES - 1) % 3]));
        } else if (MODE == 4) {  // like 2, with the wave's priority raised while it feeds the matrix pipe
            v16f cur = mfma(a[0][1], b1, mfma(a[0][0], b0, z));
#pragma unroll
            for (int t = 1; t < TILES; ++t) {
                __builtin_amdgcn_s_setprio(2);
                v16f nxt = mfma(a[t][0], b0, z);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_s_setprio(2);
                nxt = mfma(a[t][1], b1, nxt);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                mm = max(mm, max16(cur));
                cur = nxt;
            }
            mm = max(mm, max16(cur));
        } else if (MODE == 5) {  // 1 MFMA : 2 VALU pattern over a pair of tiles
            v16f cur = mfma(a[0][1], b1, mfma(a[0][0], b0, z));
#pragma unroll
            for (int t = 1; t < TILES; ++t) {
                v16f nxt = mfma(a[t][0], b0, z);
                nxt = mfma(a[t][1], b1, nxt);
                mm = max(mm, max16(cur));
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                cur = nxt;
            }
            mm = max(mm, max16(cur));
        } else if (MODE == 1) {
            v16f cur = mfma(a[0][1], b1, mfma(a[0][0], b0, z));
#pragma unroll
            for (int t = 1; t < TILES; ++t) {
                const v16f nxt = mfma(a[t][1], b1, mfma(a[t][0], b0, z));
                mm = max(mm, max16(cur));
                cur = nxt;
            }
            mm = max(mm, max16(cur));
        } else {
            v16f cur = mfma(a[0][1], b1, mfma(a[0][0], b0, z));
#pragma unroll
            for (int t = 1; t < TILES; ++t) {
                v16f nxt = mfma(a[t][0], b0, z);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // 4 VALU
                nxt = mfma(a[t][1], b1, nxt);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                mm = max(mm, max16(cur));
                cur = nxt;
            }
            mm = max(mm, max16(cur));
        }
        if (EXTRA >= 1) {
            if (__builtin_expect(__any(mm >= 0x7f000000), 0)) out[threadIdx.x] = it;  // never true
            if (EXTRA >= 2 && (it & 3) == 3) __syncthreads();
        } else {
            b0.x += mm & 1;  // keep the loop from being hoisted
        }
    }
    int r = mm;
#pragma unroll
    for (int t = 0; t < TILES; ++t) r ^= __float_as_int(keep[t]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE, int EXTRA = 0>
int run(const char* name, int cus, int* out) {
    for (int wg_per_cu = 1; wg_per_cu <= 3; ++wg_per_cu) {
        const int blocks = cus * wg_per_cu;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_loop<MODE, EXTRA>), dim3(blocks), dim3(256), 0, 0, out, 3);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double mfmas = (double)ITERS * TILES * 2 * wg_per_cu;  // per SIMD (one wave of each workgroup per SIMD)
        const double clk = best * 1e-3 * 2.4e9;
        printf("%-44s %d waves/SIMD: %8.3f ms  %5.1f clk per MFMA per SIMD  => %4.1f %% of the nominal pipe rate (32 clk @2.4 GHz)\n",
               name, wg_per_cu, best, clk / mfmas, 100.0 * 32.0 * mfmas / clk);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    int* out;
    CHK(hipMalloc(&out, (size_t)prop.multiProcessorCount * 3 * 256 * 4));
    run<0>("MFMA only", prop.multiProcessorCount, out);
    run<1>("MFMA + max trees (two accumulator sets)", prop.multiProcessorCount, out);
    run<2>("MFMA + max trees, explicit 1 MFMA : 4 VALU", prop.multiProcessorCount, out);
    run<3>("three accumulator sets, tree two tiles behind", prop.multiProcessorCount, out);
    run<4>("1 MFMA : 4 VALU with s_setprio around MFMAs", prop.multiProcessorCount, out);
    run<5>("1 MFMA : 2 VALU : 1 MFMA : 6 VALU", prop.multiProcessorCount, out);
    run<1, 1>("kernel form + B from LDS + any-check", prop.multiProcessorCount, out);
    run<1, 2>("kernel form + LDS + any-check + barrier/4 panels", prop.multiProcessorCount, out);
    run<4, 1>("setprio form + B from LDS + any-check", prop.multiProcessorCount, out);
    run<4, 2>("setprio form + LDS + any-check + barrier/4 panels", prop.multiProcessorCount, out);
    return 0;
}
