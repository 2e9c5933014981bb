// ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (developer tool).
// Measures lane-ops per clock per CU for the instructions the Hamming / PDQ kernels
// are built from, so that DESIGN.md can state the true (VALU) ceiling.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;  // independent dependency chains per lane

#define BODY(NAME, ASM_LINE)                                                            \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {         \
        uint32_t v[CHAINS];                                                             \
        uint32_t s = __builtin_amdgcn_readfirstlane(seed);                              \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x * 31 + c + seed; \
        for (int it = 0; it < ITERS; ++it) {                                            \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                             \
                _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { ASM_LINE; }        \
            }                                                                           \
        }                                                                               \
        uint32_t r = 0;                                                                 \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) r ^= v[c];                   \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                        \
    }

BODY(k_xor_vv, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_xor_sv, asm volatile("v_xor_b32 %0, %1, %0" : "+v"(v[c]) : "s"(s)))
BODY(k_bcnt, asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_bcnt0, asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(v[c])))
BODY(k_add, asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_add3, asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_min3, asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_bfi, asm volatile("v_bfi_b32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_mulf, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_addf, asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_fmaf, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_mul_sv, asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[c]) : "s"(s)))
BODY(k_sad, asm volatile("v_sad_u8 %0, %0, %1, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_dot4, asm volatile("v_dot4_i32_i8 %0, %0, %1, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_dot8, asm volatile("v_dot8_i32_i4 %0, %0, %1, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY(k_cmp, asm volatile("v_cmp_le_u32 vcc, %0, %1" : : "v"(v[c]), "v"(v[(c + 1) % CHAINS]) : "vcc"))

// packed f32: operate on register pairs
#define BODY2(NAME, ASM_LINE)                                                           \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {         \
        typedef float f2 __attribute__((ext_vector_type(2)));                           \
        f2 v[CHAINS];                                                                   \
        f2 sp; sp.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(seed | 0x3f800000u)); \
        sp.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((seed << 3) | 0x3f800000u)); \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { v[c].x = threadIdx.x + c + seed; v[c].y = c; } \
        for (int it = 0; it < ITERS; ++it) {                                            \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                             \
                _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) { ASM_LINE; }        \
            }                                                                           \
        }                                                                               \
        float r = 0;                                                                    \
        _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) r += v[c].x + v[c].y;        \
        out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(r);                       \
    }
BODY2(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY2(k_pk_add, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
BODY2(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CHAINS])))
// packed multiply with the other operand pair in SGPRs (what a DCT row held in scalar registers would use)
BODY2(k_pk_mul_sv, asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(v[c]) : "s"(sp)))
BODY2(k_pk_mul_sv_lo, asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(v[c]) : "s"(sp)))

typedef void (*kern_t)(uint32_t*, uint32_t);

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz
    printf("device %s, %d CUs, %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    uint32_t* out;
    const int blocks = cus * 8;  // 8 workgroups of 4 waves per CU = 8 waves/SIMD
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    struct { const char* name; kern_t k; int lanes_mult; } ks[] = {
        {"v_xor_b32 v,v", k_xor_vv, 1}, {"v_xor_b32 s,v", k_xor_sv, 1}, {"v_bcnt_u32_b32 v,v,v", k_bcnt, 1},
        {"v_bcnt_u32_b32 v,v,0", k_bcnt0, 1}, {"v_add_u32", k_add, 1}, {"v_add3_u32", k_add3, 1},
        {"v_min3_u32", k_min3, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_mul_f32", k_mulf, 1}, {"v_add_f32", k_addf, 1},
        {"v_fma_f32", k_fmaf, 1}, {"v_mul_f32 s,v", k_mul_sv, 1}, {"v_sad_u8", k_sad, 1},
        {"v_dot4_i32_i8", k_dot4, 1}, {"v_dot8_i32_i4", k_dot8, 1}, {"v_cmp_le_u32", k_cmp, 1},
        {"v_pk_mul_f32", k_pk_mul, 1}, {"v_pk_add_f32", k_pk_add, 1}, {"v_pk_fma_f32", k_pk_fma, 1},
        {"v_pk_mul_f32 s[2],v[2]", k_pk_mul_sv, 1}, {"v_pk_mul_f32 s[2],v[2] opsel", k_pk_mul_sv_lo, 1},
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (auto& k : ks) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 7u);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double wave_instr = (double)blocks * 4 * ITERS * 4 * CHAINS;
        const double per_s = wave_instr * 64 / (best * 1e-3);
        printf("%-24s %8.3f ms  %7.2f Tlane-op/s  %6.1f lanes/clk/CU @%.0fMHz  (%.2f cyc/wave-instr/SIMD)\n", k.name,
               best, per_s / 1e12, per_s / cus / clk, clk / 1e6, 4.0 * 64 / (per_s / cus / clk));
    }
    return 0;
}
