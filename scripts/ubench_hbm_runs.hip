// ubench_hbm_runs.hip -- achievable HBM read bandwidth on gfx950 as a function of the contiguous run length,
// for the access shape of the 512x512 down-sampler: a wave reads `rows` segments of `run` bytes, one per frame
// row (pitch 1536 B), as consecutive 16-byte pieces, one wave per 786432-byte frame, grid = resident waves.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_hbm_runs.hip -o /tmp/ubench_hbm && /tmp/ubench_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kPitch = 1536, kFrame = 512 * 1536;

// RUN bytes per row per unit, ROWS rows per unit; the wave walks its frame unit by unit (row blocks outer,
// runs inner), like the kernel walks tile rows and steps. DEPTH units are kept in flight.
template <int RUN, int ROWS, int DEPTH>
__global__ __launch_bounds__(64) void k_read(const uint8_t* __restrict__ frames, long long n, uint32_t* __restrict__ out) {
    constexpr int PPR = RUN / 16, NI = ROWS * PPR / 64;
    static_assert(ROWS * PPR % 64 == 0, "whole instructions");
    const int lane = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (long long f = blockIdx.x; f < n; f += gridDim.x) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(frames + (size_t)f * kFrame), 0, kFrame, 0x00020000);
        constexpr int UNITS = (512 / ROWS) * (kPitch / RUN);
        u32x4 v[DEPTH][NI];
#pragma unroll 1
        for (int u0 = 0; u0 < UNITS; u0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int u = u0 + d;
                const uint32_t soff = (uint32_t)(u / (kPitch / RUN)) * ROWS * kPitch + (uint32_t)(u % (kPitch / RUN)) * RUN;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const uint32_t g = 64u * i + lane;
                    v[d][i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((g / PPR) * kPitch + (g % PPR) * 16), (int)soff, 0);
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int i = 0; i < NI; ++i) acc ^= v[d][i];
        }
    }
    out[blockIdx.x * 64 + lane] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int RUN, int ROWS, int DEPTH>
int run(const uint8_t* frames, long long n, uint32_t* out, int grid) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_read<RUN, ROWS, DEPTH>), dim3(grid), dim3(64), 0, 0, frames, n, out);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    printf("run %4d B x %2d rows, %d unit(s) in flight (%5.1f KB/wave), grid %5d: %7.3f ms  %7.1f GB/s\n", RUN, ROWS, DEPTH,
           RUN * ROWS * DEPTH / 1024.0, grid, best, (double)n * kFrame / (best * 1e-3) / 1e9);
    return 0;
}

int main() {
    const long long n = 12288;
    uint8_t* frames; uint32_t* out;
    CHK(hipMalloc(&frames, (size_t)n * kFrame));
    CHK(hipMemset(frames, 1, (size_t)n * kFrame));
    CHK(hipMalloc(&out, 8192 * 64 * 4));
    for (int grid : {3072, 6144}) {
        run<96, 32, 1>(frames, n, out, grid);
        run<192, 32, 1>(frames, n, out, grid);
        run<192, 32, 2>(frames, n, out, grid);
        run<384, 32, 1>(frames, n, out, grid);
        run<768, 32, 1>(frames, n, out, grid);
        run<1536, 32, 1>(frames, n, out, grid);
        run<1536, 8, 1>(frames, n, out, grid);
        run<192, 64, 1>(frames, n, out, grid);
        run<384, 64, 1>(frames, n, out, grid);
    }
    return 0;
}
