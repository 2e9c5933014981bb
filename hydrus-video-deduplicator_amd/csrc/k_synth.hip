// k_synth.hip -- synthetic video frames generated in HBM (workload generator of BASELINE config 5).
//
// Not part of the hashing path: the end-to-end configuration is 50 000 videos x 64 frames of 64x64 gray = 13.1 GB
// of DISTINCT frames, which neither a test nor the benchmark can afford to synthesise with numpy and push over
// PCIe. Every frame is a pure function of (seed, video, frame index), so any subset can be regenerated or read
// back and handed to the CPU oracle. Content follows SURVEY.md 8d: a smooth field (8 random low-frequency
// cosines) plus uniform noise, ~5 % exact constants (quality 0 => dropped by the quality filter), and planted
// near-copies: video d with copy_of[d] = s >= 0 is video s with per-pixel noise of +-2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hvd_kernels.h"

namespace {

__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }

// one workgroup per frame, 256 lanes x 16 pixels
__global__ __launch_bounds__(256) void k_synth_frames64(uint8_t* __restrict__ out, long long v0, uint32_t fpv,
                                                        unsigned long long n_frames, uint64_t seed,
                                                        const int32_t* __restrict__ copy_of) {
    __shared__ float prm[8][4];  // fx, fy, phase, amplitude
    __shared__ float misc[3];   // noise amplitude, constant flag, constant value
    const unsigned long long fl = blockIdx.x;  // frame inside this call
    if (fl >= n_frames) return;
    const long long v = v0 + (long long)(fl / fpv);
    const uint32_t k = (uint32_t)(fl % fpv);
    const int32_t src = copy_of ? copy_of[v] : -1;
    const long long vc = src >= 0 ? (long long)src : v;  // content comes from the source video
    const uint64_t fkey = splitmix(seed ^ splitmix((uint64_t)vc * 0x100000001B3ull + k));
    if (threadIdx.x < 8) {
        const uint64_t h0 = splitmix(fkey + 4u * threadIdx.x);
        prm[threadIdx.x][0] = 4.0f * u01(h0);
        prm[threadIdx.x][1] = 4.0f * u01(splitmix(h0 + 1));
        prm[threadIdx.x][2] = 6.2831853f * u01(splitmix(h0 + 2));
        const uint64_t ha = splitmix(fkey + 99u);
        const float scale = (ha & 3u) == 0u ? 0.2f : 1.0f;  // a quarter of the frames are low-contrast
        prm[threadIdx.x][3] = (5.0f + 35.0f * u01(splitmix(h0 + 3))) * scale;
    }
    if (threadIdx.x == 8) {
        const uint64_t hn = splitmix(fkey + 777u);
        const float amps[4] = {1.0f, 2.0f, 4.0f, 16.0f};
        misc[0] = amps[hn & 3u];
        misc[1] = u01(splitmix(hn)) < 0.05f ? 1.0f : 0.0f;
        misc[2] = (float)((hn >> 8) & 255u);
    }
    __syncthreads();
    uint8_t* dst = out + fl * 4096ull;
    const uint64_t nkey = splitmix(seed * 31u + (uint64_t)v * 0x9E3779B1ull + k);  // the copy's own noise stream
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const uint32_t p0 = (uint32_t)q * 1024u + threadIdx.x * 4u;
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t p = p0 + e;
            const float y = (float)(p >> 6) * (1.0f / 64.0f), x = (float)(p & 63u) * (1.0f / 64.0f);
            float val = 128.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) val += prm[c][3] * __cosf(6.2831853f * (prm[c][0] * x + prm[c][1] * y) + prm[c][2]);
            val += (2.0f * u01(splitmix(fkey * 3u + p)) - 1.0f) * misc[0];
            if (misc[1] != 0.0f) val = misc[2];
            float pix = fminf(fmaxf(floorf(val), 0.0f), 255.0f);
            if (src >= 0) {  // planted near-copy: +-2 per pixel on top of the source frame
                const int d = (int)(splitmix(nkey + p) % 5u) - 2;
                pix = fminf(fmaxf(pix + (float)d, 0.0f), 255.0f);
            }
            packed |= (uint32_t)pix << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(dst + p0) = packed;
    }
}

}  // namespace

namespace hvd {

hipError_t launch_synth_frames64(uint8_t* d_out, long long v0, uint32_t frames_per_video, unsigned long long n_frames,
                                 uint64_t seed, const int32_t* d_copy_of, hipStream_t s) {
    if (n_frames == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_frames64, dim3((unsigned)n_frames), dim3(256), 0, s, d_out, v0, frames_per_video, n_frames,
                       seed, d_copy_of);
    return hipGetLastError();
}

}  // namespace hvd
