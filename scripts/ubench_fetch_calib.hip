// ubench_fetch_calib.hip -- what do FETCH_SIZE / WRITE_SIZE report for the access shapes of k_down512w?
// MI355X_MICROARCH.md (HBM): "On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// Other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".
// Every kernel below touches a KNOWN number of bytes exactly once (buffers far larger than the 256 MiB Infinity Cache),
// so counter / known = the factor for that shape. Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE,
// TCC_EA0_RDREQ_sum / _32B_sum, TCC_HIT_sum TCC_MISS_sum in passes of their own): scripts/profile_fetch_calib.sh.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_fetch_calib.hip -o scripts/ubench_fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kPitch = 1536, kFrame = 512 * 1536;

// (1) the calibrated case of the guide: every lane 16 bytes, lanes contiguous, the buffer streamed once
__global__ __launch_bounds__(256) void k_cal_stream(const u32x4* __restrict__ in, size_t n16, uint32_t* __restrict__ out) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc ^= in[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// (2) the down-sampler's shape: one wave per frame; a unit = ROWS rows x RUN bytes at the 1536-byte row pitch, read as
// consecutive 16-byte pieces (k_down512w: RUN = 192, ROWS = 32, six b128 loads). SPLIT = 1: the two 32-row halves of a
// 64-row tile row alternate and walk along the rows, as the kernel's lower / upper half do (neighbouring units of one
// half, which share a 128-byte line when RUN = 192, are then two units apart); SPLIT = 0: units row-block-major.
template <int RUN, int ROWS, int SPLIT>
__global__ __launch_bounds__(64) void k_cal_runs(const uint8_t* __restrict__ frames, long long n, uint32_t* __restrict__ out) {
    constexpr int PPR = RUN / 16, NI = ROWS * PPR / 64, UPR = kPitch / RUN;
    static_assert(ROWS * PPR % 64 == 0, "whole instructions");
    const int lane = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (long long f = blockIdx.x; f < n; f += gridDim.x) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(frames + (size_t)f * kFrame), 0, kFrame, 0x00020000);
        constexpr int UNITS = (512 / ROWS) * UPR;
#pragma unroll 1
        for (int u = 0; u < UNITS; ++u) {
            int rb, cu;
            if (SPLIT) {  // pairs of row blocks (2t, 2t+1) interleaved: u -> (t, cu, half)
                const int t = u / (2 * UPR), w = u % (2 * UPR);
                rb = 2 * t + (w & 1);
                cu = w >> 1;
            } else {
                rb = u / UPR;
                cu = u % UPR;
            }
            const uint32_t soff = (uint32_t)rb * ROWS * kPitch + (uint32_t)cu * RUN;
            u32x4 v[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t g = 64u * i + lane;
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((g / PPR) * kPitch + (g % PPR) * 16), (int)soff, 0);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) acc ^= v[i];
        }
    }
    out[blockIdx.x * 64 + lane] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// (3) stores: W = 1 lane-contiguous dwords (the kernel's buffer_store_dword of results and pass-B state), W = 4 b128
template <int W>
__global__ __launch_bounds__(256) void k_cal_store(uint32_t* __restrict__ outp, size_t nwords) {
    const size_t step = (size_t)gridDim.x * 256 * W;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * W; i + W <= nwords; i += step) {
        if (W == 4) {
            u32x4 v = {(uint32_t)i, 1u, 2u, 3u};
            *reinterpret_cast<u32x4*>(outp + i) = v;
        } else {
            outp[i] = (uint32_t)i;
        }
    }
}

// (4) the pass-B state's shape: every wave rewrites and re-reads its OWN small scratch (kWords dwords) `rounds` times;
// algorithmic bytes to memory: none have to leave the chip before the kernel ends (3072 waves x 10.6 KiB = 33 MB).
__global__ __launch_bounds__(64) void k_cal_scratch(uint32_t* __restrict__ scratch, int words, int rounds, uint32_t* __restrict__ out) {
    uint32_t* mine = scratch + (size_t)blockIdx.x * words;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r)
        for (int i = threadIdx.x; i < words; i += 64) {
            acc += mine[i];
            mine[i] = acc + r;
        }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    const long long n = 6144;                       // frames: 4.83 GB, 18x the Infinity Cache
    const size_t bytes = (size_t)n * kFrame;
    uint8_t* frames; uint32_t* out; uint32_t* scratch;
    CHK(hipMalloc(&frames, bytes));
    CHK(hipMemset(frames, 1, bytes));
    CHK(hipMalloc(&out, 8192 * 64 * 4));
    const int words = 2720, rounds = 9;            // pass-B state: 17 x 5 x 32 floats per wave, one round per tile row
    CHK(hipMalloc(&scratch, (size_t)3072 * words * 4));
    CHK(hipMemset(scratch, 0, (size_t)3072 * words * 4));
    CHK(hipDeviceSynchronize());
    printf("known bytes per launch: stream/runs %zu, store %zu, scratch (read = written) %zu\n", bytes, bytes,
           (size_t)3072 * words * 4 * rounds);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_cal_stream, dim3(4096), dim3(256), 0, 0, (const u32x4*)frames, bytes / 16, out);
        hipLaunchKernelGGL((k_cal_runs<192, 32, 0>), dim3(3072), dim3(64), 0, 0, frames, n, out);
        hipLaunchKernelGGL((k_cal_runs<192, 32, 1>), dim3(3072), dim3(64), 0, 0, frames, n, out);
        hipLaunchKernelGGL((k_cal_runs<384, 32, 1>), dim3(3072), dim3(64), 0, 0, frames, n, out);
        hipLaunchKernelGGL((k_cal_runs<64, 32, 1>), dim3(3072), dim3(64), 0, 0, frames, n, out);
        hipLaunchKernelGGL((k_cal_runs<128, 32, 1>), dim3(3072), dim3(64), 0, 0, frames, n, out);
        hipLaunchKernelGGL((k_cal_runs<1536, 8, 0>), dim3(3072), dim3(64), 0, 0, frames, n, out);
        hipLaunchKernelGGL(k_cal_store<1>, dim3(4096), dim3(256), 0, 0, (uint32_t*)frames, bytes / 4);
        hipLaunchKernelGGL(k_cal_store<4>, dim3(4096), dim3(256), 0, 0, (uint32_t*)frames, bytes / 4);
        hipLaunchKernelGGL(k_cal_scratch, dim3(3072), dim3(64), 0, 0, scratch, words, rounds, out);
        CHK(hipDeviceSynchronize());
    }
    CHK(hipGetLastError());
    printf("done\n");
    return 0;
}
