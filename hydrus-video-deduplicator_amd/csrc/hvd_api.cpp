// hvd_api.cpp -- host side of the C-ABI declared in include/hvd_mi355x.h.
//
// One process drives one MI355X (hvd_init(device)); all kernels go to one library
// stream. No CPU fallback exists anywhere in this file: every compute entry point
// needs an initialised device and fails with HVD_ERR_STATE / HVD_ERR_NO_DEVICE
// otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "hvd_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(HVD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

#define NCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return fail(HVD_ERR_RCCL, "%s: %s", #expr, ncclGetErrorString(r_)); \
    } while (0)

struct Ctx {
    bool ready = false;
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float* d_dct = nullptr;
    float h_dct[16 * 64];
    bool comm_ready = false;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    // grow-only staging for the candidate-pair exchange (no malloc/free inside a step)
    void* x_cnt_in = nullptr;
    void* x_cnt_all = nullptr;
    void* x_send = nullptr;
    void* x_recv = nullptr;
    size_t x_send_cap = 0, x_recv_cap = 0;
    // grow-only scratch of the legacy one-pair entry (hvd_match_two): a, b, flags, result
    void* m_a = nullptr;
    void* m_b = nullptr;
    void* m_f = nullptr;
    void* m_o = nullptr;
    size_t m_a_cap = 0, m_b_cap = 0, m_f_cap = 0;
    // pinned, device-visible staging of the small-operand path of hvd_match_two: operands, then the two counters
    uint8_t* m_pin = nullptr;
    int32_t m_seq = 0;
    std::mutex m_mu;
};
Ctx g;
std::mutex g_mu;

// pdqhashing.cpp fill_dct_matrix_64_cached: float scale * double cos, rounded once.
void fill_dct(float* out) {
    const float scale = (float)std::sqrt(2.0 / 64.0);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 64; ++j)
            out[i * 64 + j] = (float)((double)scale * std::cos((M_PI / 2 / 64.0) * (i + 1) * (2 * j + 1)));
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

int need_ready() {
    if (!g.ready) return fail(HVD_ERR_STATE, "hvd_init() has not been called (no CPU fallback exists)");
    // HIP's current device is per host thread; the reference calls this path from the main thread or
    // from a QThread worker (gui/gui.py:195-237), so every entry re-asserts the bound device.
    hipError_t e = hipSetDevice(g.device);
    if (e != hipSuccess) return fail(HVD_ERR_HIP, "hipSetDevice(%d): %s", g.device, hipGetErrorString(e));
    return HVD_OK;
}

bool pair_less(const hvd_pair& x, const hvd_pair& y) { return x.i != y.i ? x.i < y.i : x.j < y.j; }

}  // namespace

static void free_exchange_buffers();
static int grow(void** p, size_t* cap, size_t need);

namespace hvd {
int api_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
const float* api_dct_device() {
    if (!g.ready) return nullptr;
    (void)hipSetDevice(g.device);
    return g.d_dct;
}
}  // namespace hvd

extern "C" {

int hvd_abi_version(void) { return HVD_ABI_VERSION; }

int hvd_last_error(char* buf, size_t len) {
    if (!buf || len == 0) return HVD_ERR_ARG;
    snprintf(buf, len, "%s", g_err);
    return HVD_OK;
}

int hvd_device_count(int* out_n) {
    if (!out_n) return fail(HVD_ERR_ARG, "out_n is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *out_n = n;
    return HVD_OK;
}

int hvd_dct_matrix(float* out) {
    if (!out) return fail(HVD_ERR_ARG, "out is NULL");
    fill_dct(out);
    return HVD_OK;
}

int hvd_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g.ready) {
        if (g.device == device) return HVD_OK;
        return fail(HVD_ERR_STATE, "already bound to device %d (one process per GPU)", g.device);
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(HVD_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    }
    if (device < 0 || device >= n) return fail(HVD_ERR_ARG, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HVD_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&g.ev0));
    HIP_TRY(hipEventCreate(&g.ev1));
    fill_dct(g.h_dct);
    HIP_TRY(hipMalloc((void**)&g.d_dct, sizeof g.h_dct));
    HIP_TRY(hipMemcpy(g.d_dct, g.h_dct, sizeof g.h_dct, hipMemcpyHostToDevice));
    if (const char* m = getenv("HVD_PDQ_DCT_MODE")) {  // same switch as hvd_set_pdq_dct_mode()
        if (!strcmp(m, "fma") || !strcmp(m, "1")) hvd::g_pdq_dct_mode = HVD_DCT_FMA;
        else if (!strcmp(m, "strict") || !strcmp(m, "0") || !*m) hvd::g_pdq_dct_mode = HVD_DCT_STRICT;
        else return fail(HVD_ERR_ARG, "HVD_PDQ_DCT_MODE=%s: expected strict or fma", m);
    }
    g.device = device;
    g.ready = true;
    return HVD_OK;
}

int hvd_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.ready) return HVD_OK;
    (void)hipStreamSynchronize(g.stream);
    if (g.comm_ready) {
        free_exchange_buffers();
        (void)ncclCommDestroy(g.comm);
        g.comm_ready = false;
    }
    (void)hipFree(g.d_dct);
    (void)hipEventDestroy(g.ev0);
    (void)hipEventDestroy(g.ev1);
    (void)hipStreamDestroy(g.stream);
    for (void** p : {&g.m_a, &g.m_b, &g.m_f, &g.m_o})
        if (*p) (void)hipFree(*p);
    if (g.m_pin) (void)hipHostFree(g.m_pin);
    g.~Ctx();
    new (&g) Ctx();
    return HVD_OK;
}

/* ------------------------------------------------------------ device API -- */

int hvd_dev_malloc(void** out_ptr, size_t bytes) {
    if (int rc = need_ready()) return rc;
    if (!out_ptr) return fail(HVD_ERR_ARG, "out_ptr is NULL");
    HIP_TRY(hipMalloc(out_ptr, bytes ? bytes : 1));
    return HVD_OK;
}

int hvd_dev_free(void* d_ptr) {
    if (int rc = need_ready()) return rc;
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return HVD_OK;
}

int hvd_dev_memset(void* d_ptr, int value, size_t bytes) {
    if (int rc = need_ready()) return rc;
    HIP_TRY(hipMemsetAsync(d_ptr, value, bytes, g.stream));
    return HVD_OK;
}

int hvd_memcpy_h2d(void* d_dst, const void* src, size_t bytes) {
    if (int rc = need_ready()) return rc;
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return HVD_OK;
}

int hvd_memcpy_d2h(void* dst, const void* d_src, size_t bytes) {
    if (int rc = need_ready()) return rc;
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return HVD_OK;
}

int hvd_dev_sync(void) {
    if (int rc = need_ready()) return rc;
    HIP_TRY(hipStreamSynchronize(g.stream));
    return HVD_OK;
}

int hvd_timer_start(void) {
    if (int rc = need_ready()) return rc;
    HIP_TRY(hipEventRecord(g.ev0, g.stream));
    return HVD_OK;
}

int hvd_timer_stop(float* out_ms) {
    if (int rc = need_ready()) return rc;
    if (!out_ms) return fail(HVD_ERR_ARG, "out_ms is NULL");
    HIP_TRY(hipEventRecord(g.ev1, g.stream));
    HIP_TRY(hipEventSynchronize(g.ev1));
    HIP_TRY(hipEventElapsedTime(out_ms, g.ev0, g.ev1));
    return HVD_OK;
}

int hvd_set_pdq_dct_mode(int mode) {
    if (mode != HVD_DCT_STRICT && mode != HVD_DCT_FMA) return fail(HVD_ERR_ARG, "unknown DCT mode %d", mode);
    hvd::g_pdq_dct_mode = mode;
    return HVD_OK;
}

int hvd_get_pdq_dct_mode(void) { return hvd::g_pdq_dct_mode; }

int hvd_debug_set(const char* key, int value) {
    if (!key) return fail(HVD_ERR_ARG, "key is NULL");
    if (strcmp(key, "pdq_dct_from_lds") == 0) {
        hvd::g_pdq_dct_from_lds = value != 0;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_luma_lut") == 0) {
        hvd::g_pdq_luma_lut = value;
        return HVD_OK;
    }
    if (strcmp(key, "fp4_code") == 0) {
        if (value != 1 && value != 2 && value != 4 && value != 6) return fail(HVD_ERR_ARG, "fp4_code must be 1, 2, 4 or 6");
        hvd::g_fp4_code = (uint32_t)value;
        return HVD_OK;
    }
    if (strcmp(key, "mfma_col_chunk_max") == 0) {
        if (value < 256 || value % 128) return fail(HVD_ERR_ARG, "mfma_col_chunk_max must be a multiple of 128, >= 256");
        hvd::g_mfma_col_chunk_max = (uint32_t)value;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_down512_wave") == 0) {
        if (value < 0 || value > 2) return fail(HVD_ERR_ARG, "pdq_down512_wave: 0 never, 1 by batch size, 2 always");
        hvd::g_pdq_down512_wave = value;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_down512_wave_grid") == 0) {
        if (value < 0) return fail(HVD_ERR_ARG, "pdq_down512_wave_grid must not be negative (0 = default)");
        hvd::g_pdq_down512_wave_grid = value;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_fused_down512") == 0) {
        hvd::g_pdq_fused_down512 = value != 0;
        return HVD_OK;
    }
    return fail(HVD_ERR_ARG, "unknown debug key %s", key);
}

int hvd_pdq_scratch_bytes(int64_t n, int h, int w, int channels, size_t* out_bytes) {
    if (!out_bytes || n < 0 || h < 64 || w < 64 || (channels != 1 && channels != 3))
        return fail(HVD_ERR_ARG, "bad frame geometry");
    if (h == 64 && w == 64 && channels == 1) {
        *out_bytes = 0;
    } else if (h == 64 && w == 64) {
        *out_bytes = sizeof(float) * 4096 * (size_t)n;
    } else {
        const size_t cnt = (size_t)(n < 1024 ? n : 1024);
        *out_bytes = sizeof(float) * (4096 * (size_t)n + cnt * hvd::pdq_downsample_ws_floats(h, w));
    }
    return HVD_OK;
}

}  // extern "C" (the helpers below have C++ linkage; hvd_stream.cpp uses them)

namespace hvd {
size_t api_scratch_bytes(int64_t n, int h, int w, int channels) {
    size_t b = 0;
    (void)hvd_pdq_scratch_bytes(n, h, w, channels, &b);
    return b;
}

// Enqueue the PDQ kernels for one batch on stream s (geometry already validated).
hipError_t api_launch_hash(const void* d_frames, int64_t n, int h, int w, int channels, void* d_scratch, void* d_hashes,
                           void* d_quality, hipStream_t s) {
    if (h == 64 && w == 64 && channels == 1)
        return launch_pdq_hash64(d_frames, 0, n, g.d_dct, (uint8_t*)d_hashes, (int32_t*)d_quality, s);
    hipError_t e;
    float* out64 = (float*)d_scratch;
    if (h == 64 && w == 64)
        e = launch_pdq_luma64_rgb((const uint8_t*)d_frames, n, out64, s);
    else
        e = launch_pdq_downsample((const uint8_t*)d_frames, n, h, w, channels, out64 + (size_t)n * 4096, out64, s);
    if (e != hipSuccess) return e;
    return launch_pdq_hash64(d_scratch, 1, n, g.d_dct, (uint8_t*)d_hashes, (int32_t*)d_quality, s);
}
}  // namespace hvd

extern "C" {

int hvd_dev_pdq_hash_frames(const void* d_frames, int64_t n, int h, int w, int channels, void* d_scratch,
                            void* d_hashes, void* d_quality) {
    if (int rc = need_ready()) return rc;
    if (n < 0 || h < 64 || w < 64 || (channels != 1 && channels != 3))
        return fail(HVD_ERR_ARG, "bad frame geometry n=%lld h=%d w=%d channels=%d (need h,w >= 64)", (long long)n, h,
                    w, channels);
    if (h > 4096 || w > 4096) return fail(HVD_ERR_ARG, "frames larger than 4096 px per side are not supported");
    if (n == 0) return HVD_OK;
    if (!d_frames || !d_hashes || !d_quality) return fail(HVD_ERR_ARG, "NULL device pointer");
    const bool need_scratch = !(h == 64 && w == 64 && channels == 1);
    if (need_scratch && !d_scratch) return fail(HVD_ERR_ARG, "d_scratch (hvd_pdq_scratch_bytes) is required unless 64x64 gray");
    HIP_TRY(hvd::api_launch_hash(d_frames, n, h, w, channels, d_scratch, d_hashes, d_quality, g.stream));
    return HVD_OK;
}

int hvd_allpairs_tile_geometry(int64_t n, int variant, uint32_t* rows_per_block, uint32_t* col_chunk) {
    if (n < 0 || n >= (1ll << 32) || !rows_per_block || !col_chunk) return fail(HVD_ERR_ARG, "bad arguments");
    if (!hvd::allpairs_geometry((uint32_t)n, variant, rows_per_block, col_chunk) &&
        !hvd::allpairs_mfma_geometry((uint32_t)n, variant, rows_per_block, col_chunk))
        return fail(HVD_ERR_ARG, "unknown kernel variant %d", variant);
    return HVD_OK;
}

int hvd_fp4_image_bytes(int64_t n, size_t* out_bytes) {
    if (n < 0 || n >= (1ll << 32) || !out_bytes) return fail(HVD_ERR_ARG, "bad arguments");
    *out_bytes = (size_t)hvd::fp4_rows_padded((uint32_t)n) * 128u;
    return HVD_OK;
}

int hvd_dev_expand_fp4(const void* d_db, int64_t n, void* d_img) {
    if (int rc = need_ready()) return rc;
    if (n < 0 || n >= (1ll << 32) || !d_img || (n > 0 && !d_db)) return fail(HVD_ERR_ARG, "bad arguments");
    HIP_TRY(hvd::launch_expand_fp4(d_db, (uint32_t)n, d_img, g.stream));
    return HVD_OK;
}

int hvd_dev_allpairs_hamming256_mfma(const void* d_db, const void* d_img, int64_t n, const void* d_group, int max_dist, int rank,
                                     int world, void* d_pairs, int64_t cap, void* d_count, int variant) {
    if (int rc = need_ready()) return rc;
    if (n < 0 || n >= (1ll << 32)) return fail(HVD_ERR_ARG, "n=%lld out of range [0,2^32)", (long long)n);
    if (max_dist < 0 || max_dist > 256) return fail(HVD_ERR_ARG, "max_dist=%d out of range [0,256]", max_dist);
    if (world < 1 || rank < 0 || rank >= world) return fail(HVD_ERR_ARG, "bad rank/world %d/%d", rank, world);
    if (cap < 0 || !d_count || (cap > 0 && !d_pairs)) return fail(HVD_ERR_ARG, "bad output buffer");
    if (n < 2) return HVD_OK;
    if (!d_img || !d_db) return fail(HVD_ERR_ARG, "d_db / d_img is NULL");
    hvd::AllPairsArgs a;
    a.d_db = d_db;
    a.n = (uint32_t)n;
    a.d_group = (const int32_t*)d_group;
    a.max_dist = (uint32_t)max_dist;
    a.rank = (uint32_t)rank;
    a.world = (uint32_t)world;
    a.d_pairs = (hvd_pair*)d_pairs;
    a.cap = (unsigned long long)cap;
    a.d_count = (unsigned long long*)d_count;
    a.variant = variant;
    a.col_chunk = 0;
    hipError_t e = hvd::launch_allpairs_mfma(a, d_img, g.stream);
    if (e != hipSuccess) return fail(HVD_ERR_HIP, "launch_allpairs_mfma(variant=%d): %s", variant, hipGetErrorString(e));
    return HVD_OK;
}

int hvd_dev_allpairs_hamming256(const void* d_db, int64_t n, const void* d_group, int max_dist, int rank, int world,
                                void* d_pairs, int64_t cap, void* d_count, int variant) {
    if (int rc = need_ready()) return rc;
    if (n < 0 || n >= (1ll << 32)) return fail(HVD_ERR_ARG, "n=%lld out of range [0,2^32)", (long long)n);
    if (max_dist < 0 || max_dist > 256) return fail(HVD_ERR_ARG, "max_dist=%d out of range [0,256]", max_dist);
    if (world < 1 || rank < 0 || rank >= world) return fail(HVD_ERR_ARG, "bad rank/world %d/%d", rank, world);
    if (cap < 0 || !d_count || (cap > 0 && !d_pairs)) return fail(HVD_ERR_ARG, "bad output buffer");
    if (n < 2) return HVD_OK;
    if (!d_db) return fail(HVD_ERR_ARG, "d_db is NULL");
    hvd::AllPairsArgs a;
    a.d_db = d_db;
    a.n = (uint32_t)n;
    a.d_group = (const int32_t*)d_group;
    a.max_dist = (uint32_t)max_dist;
    a.rank = (uint32_t)rank;
    a.world = (uint32_t)world;
    a.d_pairs = (hvd_pair*)d_pairs;
    a.cap = (unsigned long long)cap;
    a.d_count = (unsigned long long*)d_count;
    a.variant = variant;
    a.col_chunk = 0;
    hipError_t e = hvd::launch_allpairs(a, g.stream);
    if (e != hipSuccess) return fail(HVD_ERR_HIP, "launch_allpairs(variant=%d): %s", variant, hipGetErrorString(e));
    return HVD_OK;
}

int hvd_dev_cross_hamming256_mfma(const void* d_img_q, int64_t nq, const void* d_img_t, int64_t nt,
                                  const void* d_group_q, const void* d_group_t, int max_dist, int rank, int world,
                                  void* d_pairs, int64_t cap, void* d_count) {
    if (int rc = need_ready()) return rc;
    if (nq < 0 || nt < 0 || nq >= (1ll << 32) || nt >= (1ll << 32)) return fail(HVD_ERR_ARG, "set size out of range");
    if (max_dist < 0 || max_dist >= 128)
        return fail(HVD_ERR_ARG, "cross search supports max_dist in [0,127] (the reference uses 31), got %d", max_dist);
    if (world < 1 || rank < 0 || rank >= world) return fail(HVD_ERR_ARG, "bad rank/world %d/%d", rank, world);
    if (cap < 0 || !d_count || (cap > 0 && !d_pairs)) return fail(HVD_ERR_ARG, "bad output buffer");
    if ((d_group_q == nullptr) != (d_group_t == nullptr)) return fail(HVD_ERR_ARG, "pass both group maps or neither");
    if (nq == 0 || nt == 0) return HVD_OK;
    if (!d_img_q || !d_img_t) return fail(HVD_ERR_ARG, "NULL image");
    hvd::AllPairsArgs a;
    a.d_db = nullptr;
    a.n = (uint32_t)nt;
    a.d_group = (const int32_t*)d_group_q;
    a.max_dist = (uint32_t)max_dist;
    a.rank = (uint32_t)rank;
    a.world = (uint32_t)world;
    a.d_pairs = (hvd_pair*)d_pairs;
    a.cap = (unsigned long long)cap;
    a.d_count = (unsigned long long*)d_count;
    a.variant = HVD_DEFAULT_VARIANT;
    a.col_chunk = 0;
    hipError_t e = hvd::launch_cross_mfma(a, d_img_q, (uint32_t)nq, d_img_t, (const int32_t*)d_group_t, g.stream);
    if (e != hipSuccess) return fail(HVD_ERR_HIP, "launch_cross_mfma: %s", hipGetErrorString(e));
    return HVD_OK;
}

/* ------------------------------------------------- host-buffer entry points -- */

static int hash_frames_host(const uint8_t* frames, int64_t n, int h, int w, int channels, uint8_t* out_hashes,
                            int32_t* out_quality) {
    if (int rc = need_ready()) return rc;
    if (n < 0 || h < 64 || w < 64) return fail(HVD_ERR_ARG, "bad frame geometry n=%lld h=%d w=%d", (long long)n, h, w);
    if (n == 0) return HVD_OK;
    if (!frames || !out_hashes || !out_quality) return fail(HVD_ERR_ARG, "NULL buffer");
    const size_t frame_bytes = (size_t)h * w * channels;
    // Batches bound the staging footprint (<= ~1 GiB of frames per batch).
    int64_t batch = (int64_t)((1ull << 30) / frame_bytes);
    if (batch < 1) batch = 1;
    if (batch > n) batch = n;
    const bool need_scratch = !(h == 64 && w == 64 && channels == 1);
    DevBuf d_in, d_scr, d_h, d_q;
    HIP_TRY(d_in.alloc(frame_bytes * batch));
    if (need_scratch) {
        size_t sb = 0;
        if (int rc = hvd_pdq_scratch_bytes(batch, h, w, channels, &sb)) return rc;
        HIP_TRY(d_scr.alloc(sb));
    }
    HIP_TRY(d_h.alloc(32 * (size_t)batch));
    HIP_TRY(d_q.alloc(4 * (size_t)batch));
    for (int64_t f0 = 0; f0 < n; f0 += batch) {
        const int64_t m = std::min(batch, n - f0);
        HIP_TRY(hipMemcpyAsync(d_in.p, frames + frame_bytes * f0, frame_bytes * m, hipMemcpyHostToDevice, g.stream));
        if (int rc = hvd_dev_pdq_hash_frames(d_in.p, m, h, w, channels, need_scratch ? d_scr.p : nullptr, d_h.p, d_q.p))
            return rc;
        HIP_TRY(hipMemcpyAsync(out_hashes + 32 * f0, d_h.p, 32 * (size_t)m, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipMemcpyAsync(out_quality + f0, d_q.p, 4 * (size_t)m, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
    }
    return HVD_OK;
}

int hvd_pdq_hash_frames_gray_u8(const uint8_t* frames, int64_t n, int h, int w, uint8_t* out_hashes,
                                int32_t* out_quality) {
    return hash_frames_host(frames, n, h, w, 1, out_hashes, out_quality);
}

int hvd_pdq_hash_frames_rgb24_u8(const uint8_t* frames, int64_t n, int h, int w, uint8_t* out_hashes,
                                 int32_t* out_quality) {
    return hash_frames_host(frames, n, h, w, 3, out_hashes, out_quality);
}

// Runs the default all-pairs kernel (FP4-MFMA form) on a host DB; fetches up to `cap`
// unordered records.
static int allpairs_host_raw(const uint8_t* db, int64_t n, const int32_t* group, int max_dist,
                             std::vector<hvd_pair>& recs, int64_t cap, int64_t* out_count) {
    DevBuf d_db, d_img, d_grp, d_pairs, d_cnt;
    HIP_TRY(d_db.alloc(32 * (size_t)n));
    HIP_TRY(hipMemcpyAsync(d_db.p, db, 32 * (size_t)n, hipMemcpyHostToDevice, g.stream));
    size_t img_bytes = 0;
    if (int rc = hvd_fp4_image_bytes(n, &img_bytes)) return rc;
    HIP_TRY(d_img.alloc(img_bytes));
    if (int rc = hvd_dev_expand_fp4(d_db.p, n, d_img.p)) return rc;
    if (group) {
        HIP_TRY(d_grp.alloc(4 * (size_t)n));
        HIP_TRY(hipMemcpyAsync(d_grp.p, group, 4 * (size_t)n, hipMemcpyHostToDevice, g.stream));
    }
    HIP_TRY(d_pairs.alloc(sizeof(hvd_pair) * (size_t)cap));
    HIP_TRY(d_cnt.alloc(8));
    HIP_TRY(hipMemsetAsync(d_cnt.p, 0, 8, g.stream));
    if (int rc = hvd_dev_allpairs_hamming256_mfma(d_db.p, d_img.p, n, group ? d_grp.p : nullptr, max_dist, 0, 1,
                                                  d_pairs.p, cap, d_cnt.p, HVD_DEFAULT_VARIANT))
        return rc;
    unsigned long long cnt = 0;
    HIP_TRY(hipMemcpyAsync(&cnt, d_cnt.p, 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    *out_count = (int64_t)cnt;
    const size_t m = (size_t)std::min<unsigned long long>(cnt, (unsigned long long)cap);
    recs.resize(m);
    if (m) {
        HIP_TRY(hipMemcpyAsync(recs.data(), d_pairs.p, sizeof(hvd_pair) * m, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
    }
    return HVD_OK;
}

int hvd_allpairs_hamming256(const uint8_t* db, int64_t n, const int32_t* group, int max_dist, hvd_pair* out,
                            int64_t cap, int64_t* out_count) {
    if (int rc = need_ready()) return rc;
    if (n < 0 || n >= (1ll << 32) || cap < 0 || !out_count || (cap > 0 && !out))
        return fail(HVD_ERR_ARG, "bad arguments n=%lld cap=%lld", (long long)n, (long long)cap);
    if (max_dist < 0 || max_dist > 256) return fail(HVD_ERR_ARG, "max_dist=%d out of range [0,256]", max_dist);
    *out_count = 0;
    if (n < 2) return HVD_OK;
    if (!db) return fail(HVD_ERR_ARG, "db is NULL");
    std::vector<hvd_pair> recs;
    if (int rc = allpairs_host_raw(db, n, group, max_dist, recs, cap, out_count)) return rc;
    if (*out_count > cap)
        return fail(HVD_ERR_OVERFLOW, "pair buffer too small: need %lld records, cap %lld", (long long)*out_count,
                    (long long)cap);
    std::sort(recs.begin(), recs.end(), pair_less);
    if (!recs.empty()) memcpy(out, recs.data(), sizeof(hvd_pair) * recs.size());
    return HVD_OK;
}

// Frame-level hits -> per video pair (a = video of the row frame, b = video of the column frame):
// q_hits = distinct row frames, t_hits = distinct column frames. Output sorted by (a, b).
static void aggregate_video_hits(const std::vector<hvd_pair>& recs, const int32_t* vid_row, const int32_t* vid_col,
                                 std::vector<hvd_vmatch>& res) {
    struct Key {
        uint32_t a, b, f;
    };
    std::vector<Key> qs(recs.size()), ts(recs.size());
    for (size_t k = 0; k < recs.size(); ++k) {
        const uint32_t va = (uint32_t)vid_row[recs[k].i], vb = (uint32_t)vid_col[recs[k].j];
        qs[k] = Key{va, vb, recs[k].i};
        ts[k] = Key{va, vb, recs[k].j};
    }
    auto less = [](const Key& x, const Key& y) {
        if (x.a != y.a) return x.a < y.a;
        if (x.b != y.b) return x.b < y.b;
        return x.f < y.f;
    };
    std::sort(qs.begin(), qs.end(), less);
    std::sort(ts.begin(), ts.end(), less);
    res.clear();
    size_t qi = 0, ti = 0;
    while (qi < qs.size()) {
        const uint32_t a = qs[qi].a, b = qs[qi].b;
        uint32_t qh = 0, th = 0;
        for (uint32_t last = 0xFFFFFFFFu; qi < qs.size() && qs[qi].a == a && qs[qi].b == b; ++qi)
            if (qs[qi].f != last) {
                last = qs[qi].f;
                ++qh;
            }
        for (uint32_t last = 0xFFFFFFFFu; ti < ts.size() && ts[ti].a == a && ts[ti].b == b; ++ti)
            if (ts[ti].f != last) {
                last = ts[ti].f;
                ++th;
            }
        res.push_back(hvd_vmatch{a, b, qh, th});
    }
}

int hvd_match_two(const uint8_t* a, int64_t na, const uint8_t* b, int64_t nb, int max_dist, int32_t* q_hits,
                  int32_t* t_hits) {
    if (int rc = need_ready()) return rc;
    if (na < 0 || nb < 0 || !q_hits || !t_hits || na >= (1ll << 31) || nb >= (1ll << 31))
        return fail(HVD_ERR_ARG, "bad arguments");
    if (max_dist < 0 || max_dist > 256) return fail(HVD_ERR_ARG, "max_dist=%d out of range [0,256]", max_dist);
    *q_hits = 0;
    *t_hits = 0;
    if (na == 0 || nb == 0) return HVD_OK;  // either side empty => no match (db/DedupeDB.py:555-557)
    if (!a || !b) return fail(HVD_ERR_ARG, "NULL hash buffer");
    // The VP-tree issues one such call per visited node (db/vptree.py:737): no malloc/free per call.
    std::lock_guard<std::mutex> lk(g.m_mu);
    const size_t small = hvd::match_two_small_limit();
    if (40 * (size_t)(na + nb) <= small) {
        // operands fit in LDS: the kernel reads them from pinned host memory and writes the counters there, then a
        // sequence word the host polls (a stream synchronisation costs more than the whole kernel)
        if (!g.m_pin) {
            HIP_TRY(hipHostMalloc((void**)&g.m_pin, small + 64, hipHostMallocDefault));
            memset(g.m_pin + small, 0, 64);
        }
        uint8_t* pb = g.m_pin + 32 * (size_t)na;
        volatile int32_t* ph = reinterpret_cast<volatile int32_t*>(g.m_pin + small);
        memcpy(g.m_pin, a, 32 * (size_t)na);
        memcpy(pb, b, 32 * (size_t)nb);
        const int32_t seq = ++g.m_seq == 0 ? ++g.m_seq : g.m_seq;
        HIP_TRY(hvd::launch_match_two_small((const uint32_t*)g.m_pin, (uint32_t)na, (const uint32_t*)pb, (uint32_t)nb,
                                            (uint32_t)max_dist, (int32_t*)(g.m_pin + small), seq, g.stream));
        bool seen = false;
        for (long spin = 0; spin < 4000000; ++spin) {  // ~ms; a failed launch never writes the word
            if (__atomic_load_n(&ph[2], __ATOMIC_ACQUIRE) == seq) {
                seen = true;
                break;
            }
        }
        if (!seen) {
            HIP_TRY(hipStreamSynchronize(g.stream));
            if (__atomic_load_n(&ph[2], __ATOMIC_ACQUIRE) != seq) return fail(HVD_ERR_HIP, "match kernel did not complete");
        }
        *q_hits = ph[0];
        *t_hits = ph[1];
        return HVD_OK;
    }
    if (int rc = grow(&g.m_a, &g.m_a_cap, 32 * (size_t)na)) return rc;
    if (int rc = grow(&g.m_b, &g.m_b_cap, 32 * (size_t)nb)) return rc;
    if (int rc = grow(&g.m_f, &g.m_f_cap, 4 * (size_t)nb)) return rc;
    if (!g.m_o) HIP_TRY(hipMalloc(&g.m_o, 8));
    HIP_TRY(hipMemcpyAsync(g.m_a, a, 32 * (size_t)na, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(g.m_b, b, 32 * (size_t)nb, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hvd::launch_match_two((const uint32_t*)g.m_a, (uint32_t)na, (const uint32_t*)g.m_b, (uint32_t)nb,
                                  (uint32_t)max_dist, (uint32_t*)g.m_f, (int32_t*)g.m_o, g.stream));
    int32_t hits[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(hits, g.m_o, 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    *q_hits = hits[0];
    *t_hits = hits[1];
    return HVD_OK;
}

int hvd_vpdq_match_videos(const uint8_t* frames, const int64_t* offsets, int64_t V, int max_dist, hvd_vmatch* out,
                          int64_t cap, int64_t* out_count) {
    if (int rc = need_ready()) return rc;
    if (V < 0 || !offsets || !out_count || cap < 0 || (cap > 0 && !out)) return fail(HVD_ERR_ARG, "bad arguments");
    if (max_dist < 0 || max_dist > 256) return fail(HVD_ERR_ARG, "max_dist=%d out of range [0,256]", max_dist);
    *out_count = 0;
    if (offsets[0] != 0) return fail(HVD_ERR_ARG, "offsets[0] must be 0");
    for (int64_t v = 0; v < V; ++v)
        if (offsets[v + 1] < offsets[v]) return fail(HVD_ERR_ARG, "offsets must be non-decreasing");
    const int64_t nf = V > 0 ? offsets[V] : 0;
    if (nf >= (1ll << 32) || V >= (1ll << 31)) return fail(HVD_ERR_ARG, "too many frames/videos");
    if (nf < 2) return HVD_OK;
    if (!frames) return fail(HVD_ERR_ARG, "frames is NULL");
    // frame -> video map; the kernel drops hits inside one video.
    std::vector<int32_t> vid((size_t)nf);
    for (int64_t v = 0; v < V; ++v)
        for (int64_t f = offsets[v]; f < offsets[v + 1]; ++f) vid[(size_t)f] = (int32_t)v;
    // Frame-level hits; grow the buffer until they fit.
    std::vector<hvd_pair> recs;
    int64_t fcap = std::max<int64_t>(1 << 16, nf), fcount = 0;
    for (;;) {
        if (int rc = allpairs_host_raw(frames, nf, vid.data(), max_dist, recs, fcap, &fcount)) return rc;
        if (fcount <= fcap) break;
        fcap = fcount;
    }
    std::vector<hvd_vmatch> res;
    aggregate_video_hits(recs, vid.data(), vid.data(), res);
    *out_count = (int64_t)res.size();
    if ((int64_t)res.size() > cap)
        return fail(HVD_ERR_OVERFLOW, "video match buffer too small: need %lld, cap %lld", (long long)res.size(),
                    (long long)cap);
    if (!res.empty()) memcpy(out, res.data(), sizeof(hvd_vmatch) * res.size());
    return HVD_OK;
}

static int check_offsets(const int64_t* offsets, int64_t V, int64_t* nf) {
    if (V < 0 || !offsets) return fail(HVD_ERR_ARG, "bad offsets");
    if (offsets[0] != 0) return fail(HVD_ERR_ARG, "offsets[0] must be 0");
    for (int64_t v = 0; v < V; ++v)
        if (offsets[v + 1] < offsets[v]) return fail(HVD_ERR_ARG, "offsets must be non-decreasing");
    *nf = V > 0 ? offsets[V] : 0;
    if (*nf >= (1ll << 32) || V >= (1ll << 31)) return fail(HVD_ERR_ARG, "too many frames/videos");
    return HVD_OK;
}

int hvd_vpdq_match_videos_cross(const uint8_t* frames_q, const int64_t* offsets_q, int64_t VQ, const int32_t* ids_q,
                                const uint8_t* frames_t, const int64_t* offsets_t, int64_t VT, const int32_t* ids_t,
                                int max_dist, hvd_vmatch* out, int64_t cap, int64_t* out_count) {
    if (int rc = need_ready()) return rc;
    if (!out_count || cap < 0 || (cap > 0 && !out)) return fail(HVD_ERR_ARG, "bad output buffer");
    if ((ids_q == nullptr) != (ids_t == nullptr)) return fail(HVD_ERR_ARG, "pass both id arrays or neither");
    if (max_dist < 0 || max_dist >= 128) return fail(HVD_ERR_ARG, "max_dist=%d out of range [0,127]", max_dist);
    *out_count = 0;
    int64_t nq = 0, nt = 0;
    if (int rc = check_offsets(offsets_q, VQ, &nq)) return rc;
    if (int rc = check_offsets(offsets_t, VT, &nt)) return rc;
    if (nq == 0 || nt == 0) return HVD_OK;
    if (!frames_q || !frames_t) return fail(HVD_ERR_ARG, "frames is NULL");
    std::vector<int32_t> vq((size_t)nq), vt((size_t)nt), gq, gt;
    for (int64_t v = 0; v < VQ; ++v)
        for (int64_t f = offsets_q[v]; f < offsets_q[v + 1]; ++f) vq[(size_t)f] = (int32_t)v;
    for (int64_t v = 0; v < VT; ++v)
        for (int64_t f = offsets_t[v]; f < offsets_t[v + 1]; ++f) vt[(size_t)f] = (int32_t)v;
    if (ids_q) {  // frames of videos with equal ids are not compared (a query that is also in the target set)
        gq.resize((size_t)nq);
        gt.resize((size_t)nt);
        for (int64_t f = 0; f < nq; ++f) gq[(size_t)f] = ids_q[vq[(size_t)f]];
        for (int64_t f = 0; f < nt; ++f) gt[(size_t)f] = ids_t[vt[(size_t)f]];
    }
    DevBuf d_q, d_t, d_iq, d_it, d_gq, d_gt, d_pairs, d_cnt;
    size_t bq = 0, bt = 0;
    if (int rc = hvd_fp4_image_bytes(nq, &bq)) return rc;
    if (int rc = hvd_fp4_image_bytes(nt, &bt)) return rc;
    HIP_TRY(d_q.alloc(32 * (size_t)nq));
    HIP_TRY(d_t.alloc(32 * (size_t)nt));
    HIP_TRY(d_iq.alloc(bq));
    HIP_TRY(d_it.alloc(bt));
    HIP_TRY(hipMemcpyAsync(d_q.p, frames_q, 32 * (size_t)nq, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_t.p, frames_t, 32 * (size_t)nt, hipMemcpyHostToDevice, g.stream));
    if (int rc = hvd_dev_expand_fp4(d_q.p, nq, d_iq.p)) return rc;
    if (int rc = hvd_dev_expand_fp4(d_t.p, nt, d_it.p)) return rc;
    if (ids_q) {
        HIP_TRY(d_gq.alloc(4 * (size_t)nq));
        HIP_TRY(d_gt.alloc(4 * (size_t)nt));
        HIP_TRY(hipMemcpyAsync(d_gq.p, gq.data(), 4 * (size_t)nq, hipMemcpyHostToDevice, g.stream));
        HIP_TRY(hipMemcpyAsync(d_gt.p, gt.data(), 4 * (size_t)nt, hipMemcpyHostToDevice, g.stream));
    }
    HIP_TRY(d_cnt.alloc(8));
    std::vector<hvd_pair> recs;
    int64_t fcap = std::max<int64_t>(1 << 16, nq);
    for (;;) {
        if (d_pairs.p) {
            HIP_TRY(hipFree(d_pairs.p));
            d_pairs.p = nullptr;
        }
        HIP_TRY(d_pairs.alloc(sizeof(hvd_pair) * (size_t)fcap));
        HIP_TRY(hipMemsetAsync(d_cnt.p, 0, 8, g.stream));
        if (int rc = hvd_dev_cross_hamming256_mfma(d_iq.p, nq, d_it.p, nt, ids_q ? d_gq.p : nullptr,
                                                   ids_q ? d_gt.p : nullptr, max_dist, 0, 1, d_pairs.p, fcap, d_cnt.p))
            return rc;
        unsigned long long cnt = 0;
        HIP_TRY(hipMemcpyAsync(&cnt, d_cnt.p, 8, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        if ((int64_t)cnt > fcap) {
            fcap = (int64_t)cnt;
            continue;
        }
        recs.resize((size_t)cnt);
        if (cnt) {
            HIP_TRY(hipMemcpyAsync(recs.data(), d_pairs.p, sizeof(hvd_pair) * (size_t)cnt, hipMemcpyDeviceToHost,
                                   g.stream));
            HIP_TRY(hipStreamSynchronize(g.stream));
        }
        break;
    }
    std::vector<hvd_vmatch> res;
    aggregate_video_hits(recs, vq.data(), vt.data(), res);
    *out_count = (int64_t)res.size();
    if ((int64_t)res.size() > cap)
        return fail(HVD_ERR_OVERFLOW, "video match buffer too small: need %lld, cap %lld", (long long)res.size(),
                    (long long)cap);
    if (!res.empty()) memcpy(out, res.data(), sizeof(hvd_vmatch) * res.size());
    return HVD_OK;
}

/* ------------------------------------------------------- RCCL exchange ---- */

int hvd_comm_unique_id(uint8_t out_id[HVD_UNIQUE_ID_BYTES]) {
    if (!out_id) return fail(HVD_ERR_ARG, "out_id is NULL");
    static_assert(sizeof(ncclUniqueId) <= HVD_UNIQUE_ID_BYTES, "unique id does not fit");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memset(out_id, 0, HVD_UNIQUE_ID_BYTES);
    memcpy(out_id, &id, sizeof id);
    return HVD_OK;
}

int hvd_comm_init(const uint8_t id_bytes[HVD_UNIQUE_ID_BYTES], int rank, int world) {
    if (int rc = need_ready()) return rc;
    if (!id_bytes || world < 1 || rank < 0 || rank >= world) return fail(HVD_ERR_ARG, "bad rank/world");
    if (g.comm_ready) return fail(HVD_ERR_STATE, "communicator already initialised");
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    NCCL_TRY(ncclCommInitRank(&g.comm, world, id, rank));
    g.comm_ready = true;
    g.rank = rank;
    g.world = world;
    return HVD_OK;
}

static void free_exchange_buffers() {
    void** ps[] = {&g.x_cnt_in, &g.x_cnt_all, &g.x_send, &g.x_recv};
    for (void** p : ps) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    g.x_send_cap = g.x_recv_cap = 0;
}

int hvd_comm_destroy(void) {
    if (g.comm_ready) {
        (void)hipStreamSynchronize(g.stream);
        free_exchange_buffers();
        NCCL_TRY(ncclCommDestroy(g.comm));
        g.comm_ready = false;
    }
    return HVD_OK;
}

int hvd_comm_allgather_bytes(const void* d_send, void* d_recv, size_t bytes_per_rank) {
    if (int rc = need_ready()) return rc;
    if (!g.comm_ready) return fail(HVD_ERR_STATE, "hvd_comm_init() has not been called");
    NCCL_TRY(ncclAllGather(d_send, d_recv, bytes_per_rank, ncclUint8, g.comm, g.stream));
    return HVD_OK;
}

static int grow(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return HVD_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t want = need < (1u << 16) ? (1u << 16) : need + need / 2;
    HIP_TRY(hipMalloc(p, want));
    *cap = want;
    return HVD_OK;
}

int hvd_comm_allgather_pairs(const void* d_pairs, int64_t count, hvd_pair* out_host, int64_t cap,
                             int64_t* out_total) {
    if (int rc = need_ready()) return rc;
    if (!g.comm_ready) return fail(HVD_ERR_STATE, "hvd_comm_init() has not been called");
    if (count < 0 || cap < 0 || !out_total) return fail(HVD_ERR_ARG, "bad arguments");
    const int W = g.world;
    // 1) counts
    if (!g.x_cnt_in) HIP_TRY(hipMalloc(&g.x_cnt_in, 8));
    if (!g.x_cnt_all) HIP_TRY(hipMalloc(&g.x_cnt_all, 8 * (size_t)W));
    unsigned long long c = (unsigned long long)count;
    HIP_TRY(hipMemcpyAsync(g.x_cnt_in, &c, 8, hipMemcpyHostToDevice, g.stream));
    NCCL_TRY(ncclAllGather(g.x_cnt_in, g.x_cnt_all, 1, ncclUint64, g.comm, g.stream));
    std::vector<unsigned long long> counts((size_t)W);
    HIP_TRY(hipMemcpyAsync(counts.data(), g.x_cnt_all, 8 * (size_t)W, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    unsigned long long mx = 0, total = 0;
    for (int r = 0; r < W; ++r) {
        mx = std::max(mx, counts[(size_t)r]);
        total += counts[(size_t)r];
    }
    *out_total = (int64_t)total;
    if ((int64_t)total > cap) return fail(HVD_ERR_OVERFLOW, "need %llu records, cap %lld", total, (long long)cap);
    if (total == 0) return HVD_OK;
    if (!out_host) return fail(HVD_ERR_ARG, "out_host is NULL");
    // 2) records, padded to the max count so that one all-gather suffices
    if (int rc = grow(&g.x_send, &g.x_send_cap, sizeof(hvd_pair) * (size_t)mx)) return rc;
    if (int rc = grow(&g.x_recv, &g.x_recv_cap, sizeof(hvd_pair) * (size_t)mx * (size_t)W)) return rc;
    if ((unsigned long long)count < mx)
        HIP_TRY(hipMemsetAsync((char*)g.x_send + sizeof(hvd_pair) * (size_t)count, 0,
                               sizeof(hvd_pair) * (size_t)(mx - (unsigned long long)count), g.stream));
    if (count > 0)
        HIP_TRY(hipMemcpyAsync(g.x_send, d_pairs, sizeof(hvd_pair) * (size_t)count, hipMemcpyDeviceToDevice, g.stream));
    NCCL_TRY(ncclAllGather(g.x_send, g.x_recv, sizeof(hvd_pair) * (size_t)mx, ncclUint8, g.comm, g.stream));
    std::vector<hvd_pair> all((size_t)mx * (size_t)W);
    HIP_TRY(hipMemcpyAsync(all.data(), g.x_recv, sizeof(hvd_pair) * all.size(), hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    size_t o = 0;
    for (int r = 0; r < W; ++r) {
        memcpy(out_host + o, all.data() + (size_t)r * (size_t)mx, sizeof(hvd_pair) * (size_t)counts[(size_t)r]);
        o += (size_t)counts[(size_t)r];
    }
    return HVD_OK;
}

}  // extern "C"
