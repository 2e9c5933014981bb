// hvd_fp4.h -- layout of the FP4 image shared by its writers (k_fp4_image.hip) and its reader (k_hamming_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// chunk c (16 B: bits 32c .. 32c+31 as 32 e2m1 nibbles) of hash n lives in slot c ^ ((n >> 1) & 7) of the hash's 128 bytes
__device__ __forceinline__ uint32_t img_slot(uint32_t hash, uint32_t chunk) { return chunk ^ ((hash >> 1) & 7u); }
