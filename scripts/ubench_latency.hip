// ubench_latency.hip -- dependent-issue latency of VALU / LDS on gfx950 at low occupancy (developer tool).
// The one-wave-per-frame down-sampler (k_down512w) runs long dependent chains with 2 waves per SIMD;
// this measures how many shader cycles one wave needs per instruction of such a chain, for 1..4 waves per
// SIMD and 1..4 independent chains per wave.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_latency.hip -o /tmp/ubench_lat && /tmp/ubench_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 2048, UNROLL = 16;

template <int CHAINS, int OP>
__global__ __launch_bounds__(64) void k_chain(float* out, long long* cyc, float seed) {
    __shared__ float lds[64 * 17];
    float v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x + c + seed;
    const float x = seed * 0.5f, y = seed * 0.25f;
    lds[threadIdx.x * 17] = seed;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(x));
                if (OP == 1) asm volatile("v_add_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %2" : "+v"(v[c]) : "v"(x), "v"(y));
                if (OP == 2) asm volatile("v_add_f32 %0, %0, %1\n\tv_sub_f32 %0, %0, %2\n\tv_mul_f32 %3, 0.25, %0" : "+v"(v[c]), "=v"(lds[0]) : "v"(x), "v"(y));
            }
        }
    }
    const long long t1 = clock64();
    float r = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r += v[c];
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS, int OP>
int run(const char* name, int cus, float* out, long long* cyc, int instr_per_step) {
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = cus * 4 * wps;
        hipLaunchKernelGGL((k_chain<CHAINS, OP>), dim3(blocks), dim3(64), 0, 0, out, cyc, 3.0f);  // warm
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_chain<CHAINS, OP>), dim3(blocks), dim3(64), 0, 0, out, cyc, 3.0f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        long long h[64];
        CHK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
        double avg = 0;
        for (int i = 0; i < 64; ++i) avg += h[i];
        avg /= 64;
        const double per_wave = avg / ((double)ITERS * UNROLL * CHAINS * instr_per_step);
        const double ns_per_instr_simd = ms * 1e6 / ((double)ITERS * UNROLL * CHAINS * instr_per_step * wps);
        printf("%-28s chains=%d waves/SIMD=%d : %6.2f ticks/instr/wave, %5.2f ticks/instr/SIMD | wall %.3f ns/instr/SIMD = %.2f clk @2.4GHz\n",
               name, CHAINS, wps, per_wave, per_wave / wps, ns_per_instr_simd, ns_per_instr_simd * 2.4);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs (clock64 = s_memtime ticks; 100 MHz constant clock on gfx9: scale by shader MHz / 100)\n", prop.gcnArchName, cus);
    float* out; long long* cyc;
    CHK(hipMalloc(&out, (size_t)cus * 32 * 64 * 4));
    CHK(hipMalloc(&cyc, (size_t)cus * 32 * 8));
    run<1, 0>("v_add_f32 dependent", cus, out, cyc, 1);
    run<2, 0>("v_add_f32 dependent", cus, out, cyc, 1);
    run<4, 0>("v_add_f32 dependent", cus, out, cyc, 1);
    run<1, 1>("add+sub dependent pair", cus, out, cyc, 2);
    run<2, 1>("add+sub dependent pair", cus, out, cyc, 2);
    run<8, 0>("v_add_f32 dependent", cus, out, cyc, 1);
    return 0;
}
