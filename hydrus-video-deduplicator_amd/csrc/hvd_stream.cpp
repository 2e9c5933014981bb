// hvd_stream.cpp -- streaming frame hasher behind vpdq.VideoHasher (reference
// vpdqpy/vpdqpy.py:113-119): a decoder thread pushes frames one at a time; frames are
// staged in a ring of pinned batch slots; every slot owns a HIP stream on which its batch is
// uploaded (async H2D), hashed (PDQ kernels) and its 36 bytes per frame downloaded. Uploading
// batch k+1 overlaps hashing batch k; push() blocks only when every slot is still in flight,
// which bounds the staging memory like the reference's blocking frame queue
// (vpdqpy.py:115-117).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>

#include "copy_pool.h"
#include "hvd_kernels.h"

// HVD_ABL_STREAM_NOHASH: timing-only ablation (WRONG RESULTS: batches are uploaded, nothing is hashed) -- is the upload rate of
// the pipeline held back by the kernels that run beside it? Like every ablation it needs -DHVD_DEV_ABLATION as well.
#if defined(HVD_ABL_STREAM_NOHASH) && !defined(HVD_DEV_ABLATION)
#error "HVD_ABL_STREAM_NOHASH is a developer ablation build (wrong results): add -DHVD_DEV_ABLATION to confirm"
#endif

namespace hvd {
int api_fail(int code, const char* fmt, ...);     // hvd_api.cpp
const float* api_dct_device();                    // hvd_api.cpp; nullptr before hvd_init
int api_bind_device();                            // hvd_api.cpp: hipSetDevice(bound device) for the calling thread
int api_context();                                // the calling thread's current context of the device group
void api_set_context(int idx);
size_t api_scratch_bytes(int64_t n, int h, int w, int channels);
hipError_t api_launch_hash(const void* d_frames, int64_t n, int h, int w, int channels, void* d_scratch, void* d_hashes,
                           void* d_quality, hipStream_t s);
}  // namespace hvd

namespace {

// Six slots of (by default) 32 MiB: a slot is refilled only when its previous batch has been hashed and read back, and the kernels
// of a small batch take ~0.25 ms whatever its size -- with three slots of 64 MiB the uploads of the two batches behind it did not
// always cover that (round 5: hash_frame(bytes) 16.9 -> 15.9 us per 512x512 frame, within 2 % of the zero-copy feed).
constexpr int kSlots = 6;

struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    uint8_t* h_frames = nullptr;   // pinned
    uint8_t* h_hashes = nullptr;   // pinned
    int32_t* h_quality = nullptr;  // pinned
    void* d_frames = nullptr;
    void* d_scratch = nullptr;
    void* d_hashes = nullptr;
    void* d_quality = nullptr;
    int64_t filled = 0;    // frames staged in h_frames
    int64_t in_flight = 0; // frames submitted and not yet collected
};

}  // namespace

namespace {
hvd::CopyPool g_copy_pool;  // (copy_pool.h)
// where the host side of the feed spends its time, process-wide, in nanoseconds (hvd_debug_get "hasher_ns_copy" / "_submit" /
// "_wait"; reading clears): the frame copies into the ring, the enqueue of a batch (H2D + kernels + D2H + event), the waits for
// a slot whose previous batch is still in flight. Three clock reads per frame (~60 ns).
std::atomic<long long> g_ns_copy{0}, g_ns_submit{0}, g_ns_wait{0};
struct NsScope {
    std::atomic<long long>& acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit NsScope(std::atomic<long long>& a) : acc(a) {}
    ~NsScope() { acc.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed); }
};
}  // namespace
namespace hvd {
long long stream_take_ns(int which) { return (which == 0 ? g_ns_copy : which == 1 ? g_ns_submit : g_ns_wait).exchange(0); }
}

#ifndef HVD_NO_BENCH_SYMBOLS
extern "C" int hvd_debug_parallel_copy(void* dst, const void* src, size_t n, int threads) {
    if ((!dst || !src) && n) return hvd::api_fail(HVD_ERR_ARG, "NULL buffer");
    g_copy_pool.copy((uint8_t*)dst, (const uint8_t*)src, n, threads);
    return HVD_OK;
}
#endif

// Library default of hvd_hasher_set_threads(h, 0): a quarter of the CPUs this process may run on, between 2 and 8 (the pool's
// maximum). Eight threads were never slower than four on the hosts measured and up to 15 % faster where the copy competes
// with the DMA engine for host memory (short bursts leave the memory to the engine: profiles/r05_vh_where.txt, end); a small
// machine keeps its cores for the decoder.
static int default_copy_threads() {
    static const int n = [] {
        int cpus = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) cpus = CPU_COUNT(&set);
        return std::max(2, std::min(1 + hvd::CopyPool::kMaxHelpers, cpus / 4));
    }();
    return n;
}

struct hvd_hasher {
    int ctx = 0;           // the context (device of the group) this hasher was created on: every call runs there
    int copy_threads = default_copy_threads();  // threads that share one frame's copy into the ring (hvd_hasher_set_threads)
    int w = 0, h = 0, channels = 0;
    int64_t batch = 0;             // frames a slot holds
    int64_t limit = 0;             // frames after which the CURRENT batch is submitted: ramps up to `batch` (first_limit)
    size_t frame_bytes = 0;
    Slot slot[kSlots];
    int cur = 0;
    std::vector<uint8_t> hashes;   // collected results, frame order
    std::vector<int32_t> quality;
    bool acquired = false;         // hvd_hasher_acquire handed out the next frame's slot memory
    int64_t acquired_n = 0;        // ... and hvd_hasher_acquire_n this many frames of it
};

// A video starts on an empty pipeline (the reference makes one hasher per video and finish() drains it, vpdqpy/vpdqpy.py:113-119):
// with full 64 MiB batches the DMA engine idled for the whole first batch's fill time at the start of every video -- 0.94 ms of a
// 5.5 ms video of 300 frames at 512x512 (profiles/r05_vh_where.txt). The first batch of a video is therefore SMALL (4 MiB of
// frames) and the batch size GROWS from submit to submit up to the slot's capacity; finish() starts the ramp again. The growth
// factor matters: the engine uploads batch k while the host fills batch k+1, and the host (4 copy threads: ~10 us per 786 KB
// frame) is only 1.4x faster than the link (13.7 us) -- doubling made every batch of the ramp take longer to fill than its
// predecessor took to upload (0.44 ms of idle link per video); x 1.25 keeps the link busy. Frames of 4 KB never leave the first
// step (1024 of them), so the small-frame path still submits once per video.
static int64_t first_limit(const hvd_hasher* hs) {
    const int64_t f = (int64_t)(((size_t)4 << 20) / hs->frame_bytes);
    return std::max<int64_t>(1, std::min<int64_t>(hs->batch, f));
}
static int64_t next_limit(const hvd_hasher* hs) { return std::min<int64_t>(hs->batch, hs->limit + std::max<int64_t>(1, hs->limit / 4)); }

// A hasher lives on the context it was created on, whatever context the calling thread has selected.
namespace {
struct CtxScope {
    int saved;
    explicit CtxScope(int ctx) : saved(hvd::api_context()) { hvd::api_set_context(ctx); }
    ~CtxScope() {
        hvd::api_set_context(saved);
        (void)hvd::api_bind_device();
    }
};
}  // namespace

#define S_TRY(expr)                                                                                        \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return hvd::api_fail(HVD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

static int collect(hvd_hasher* hs, Slot& s) {
    if (s.in_flight == 0) return HVD_OK;
    {
        NsScope ns(g_ns_wait);
        S_TRY(hipEventSynchronize(s.done));
    }
    hs->hashes.insert(hs->hashes.end(), s.h_hashes, s.h_hashes + 32 * s.in_flight);
    hs->quality.insert(hs->quality.end(), s.h_quality, s.h_quality + s.in_flight);
    s.in_flight = 0;
    return HVD_OK;
}

static int submit(hvd_hasher* hs, Slot& s) {
    if (s.filled == 0) return HVD_OK;
    NsScope ns(g_ns_submit);
    const int64_t m = s.filled;
    S_TRY(hipMemcpyAsync(s.d_frames, s.h_frames, hs->frame_bytes * (size_t)m, hipMemcpyHostToDevice, s.stream));
#ifndef HVD_ABL_STREAM_NOHASH
    S_TRY(hvd::api_launch_hash(s.d_frames, m, hs->h, hs->w, hs->channels, s.d_scratch, s.d_hashes, s.d_quality, s.stream));
#endif
    S_TRY(hipMemcpyAsync(s.h_hashes, s.d_hashes, 32 * (size_t)m, hipMemcpyDeviceToHost, s.stream));
    S_TRY(hipMemcpyAsync(s.h_quality, s.d_quality, 4 * (size_t)m, hipMemcpyDeviceToHost, s.stream));
    S_TRY(hipEventRecord(s.done, s.stream));
    s.in_flight = m;
    s.filled = 0;
    hs->limit = next_limit(hs);
    return HVD_OK;
}

// The reference creates one VideoHasher per video (vpdqpy/vpdqpy.py:113) and videos come strictly one after the
// other (dedup.py:346-352). Pinning and unpinning 6 x 32 MiB of host memory per video costs more than hashing a
// short video, so a destroyed hasher's slots (pinned staging, device buffers, streams, events) are PARKED and the
// next hvd_hasher_create with the same geometry takes them over. At most kMaxParked sets are kept (a GUI worker
// and the CLI never run more than one or two hashers at a time); hvd_shutdown() releases them.
namespace {
constexpr size_t kMaxParked = 2;
std::mutex g_park_mu;
std::vector<hvd_hasher*> g_parked;

void free_hasher(hvd_hasher* hs) {
    for (Slot& s : hs->slot) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.h_frames) (void)hipHostFree(s.h_frames);
        if (s.h_hashes) (void)hipHostFree(s.h_hashes);
        if (s.h_quality) (void)hipHostFree(s.h_quality);
        if (s.d_frames) (void)hipFree(s.d_frames);
        if (s.d_scratch) (void)hipFree(s.d_scratch);
        if (s.d_hashes) (void)hipFree(s.d_hashes);
        if (s.d_quality) (void)hipFree(s.d_quality);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete hs;
}
}  // namespace

namespace hvd {
void stream_set_copy_nt(int on) { set_copy_nt(on); }   // hvd_debug_set("copy_nt", 0|1)
int stream_copy_nt_level() { return copy_nt_level(); }  // hvd_debug_get("copy_nt"): 0 plain memcpy, 2 AVX2, 3 AVX-512 streaming stores
void stream_release_cache() {
    g_copy_pool.stop();
    std::lock_guard<std::mutex> lk(g_park_mu);
    for (hvd_hasher* hs : g_parked) free_hasher(hs);
    g_parked.clear();
}
}  // namespace hvd

extern "C" {

int hvd_hasher_destroy(hvd_hasher* hs) {
    if (!hs) return HVD_OK;
    CtxScope scope(hs->ctx);
    (void)hvd::api_bind_device();
    bool complete = true;
    for (Slot& s : hs->slot) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        complete = complete && s.stream && s.done && s.h_frames && s.h_hashes && s.h_quality && s.d_frames && s.d_hashes &&
                   s.d_quality;
        s.filled = s.in_flight = 0;
    }
    if (complete && hvd::api_dct_device()) {  // park it for the next video (never after hvd_shutdown)
        hs->hashes.clear();
        hs->quality.clear();
        hs->cur = 0;
        hs->acquired = false;
        hs->limit = first_limit(hs);
        std::lock_guard<std::mutex> lk(g_park_mu);
        g_parked.push_back(hs);
        if (g_parked.size() > kMaxParked) {
            free_hasher(g_parked.front());
            g_parked.erase(g_parked.begin());
        }
        return HVD_OK;
    }
    free_hasher(hs);
    return HVD_OK;
}

int hvd_hasher_create(int width, int height, int channels, int64_t batch_frames, hvd_hasher** out) {
    if (!out) return hvd::api_fail(HVD_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!hvd::api_dct_device()) return hvd::api_fail(HVD_ERR_STATE, "hvd_init() has not been called (no CPU fallback exists)");
    if (width < 64 || height < 64 || width > 4096 || height > 4096 || (channels != 1 && channels != 3) || batch_frames < 1)
        return hvd::api_fail(HVD_ERR_ARG, "bad hasher geometry %dx%dx%d batch %lld", width, height, channels,
                             (long long)batch_frames);
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        for (size_t k = g_parked.size(); k-- > 0;) {
            hvd_hasher* p = g_parked[k];
            if (p->ctx == hvd::api_context() && p->w == width && p->h == height && p->channels == channels && p->batch == batch_frames) {
                g_parked.erase(g_parked.begin() + (long)k);
                p->copy_threads = default_copy_threads();
                *out = p;
                return HVD_OK;
            }
        }
    }
    if (int rc = hvd::api_bind_device()) return rc;
    hvd_hasher* hs = new hvd_hasher();
    hs->ctx = hvd::api_context();
    hs->w = width;
    hs->h = height;
    hs->channels = channels;
    hs->batch = batch_frames;
    hs->frame_bytes = (size_t)width * height * channels;
    hs->limit = first_limit(hs);
    const size_t scratch = hvd::api_scratch_bytes(batch_frames, height, width, channels);
    for (Slot& s : hs->slot) {
        hipError_t e = hipSuccess;
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.h_frames, hs->frame_bytes * (size_t)batch_frames, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.h_hashes, 32 * (size_t)batch_frames, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.h_quality, 4 * (size_t)batch_frames, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&s.d_frames, hs->frame_bytes * (size_t)batch_frames);
        if (e == hipSuccess && scratch) e = hipMalloc(&s.d_scratch, scratch);
        if (e == hipSuccess) e = hipMalloc(&s.d_hashes, 32 * (size_t)batch_frames);
        if (e == hipSuccess) e = hipMalloc(&s.d_quality, 4 * (size_t)batch_frames);
        if (e != hipSuccess) {
            free_hasher(hs);
            hvd::stream_release_cache();  // the parked sets may be what exhausted the pinned / device memory
            return hvd::api_fail(HVD_ERR_HIP, "hasher allocation: %s", hipGetErrorString(e));
        }
    }
    *out = hs;
    return HVD_OK;
}

/* Zero-copy feed: *out_frame is where the NEXT frame (width*height*channels bytes) belongs, inside the
 * pinned batch slot, so a decoder can reformat straight into it (the reference makes a Python bytes copy
 * per frame instead, vpdqpy/vpdqpy.py:118). Blocks only when that slot's previous batch is still being
 * hashed. The frame counts once hvd_hasher_commit() is called; acquire without commit may be repeated. */
int hvd_hasher_acquire(hvd_hasher* hs, uint8_t** out_frame) {
    if (!hs || !out_frame) return hvd::api_fail(HVD_ERR_ARG, "NULL hasher/out_frame");
    CtxScope scope(hs->ctx);
    if (int rc = hvd::api_bind_device()) return rc;
    Slot& s = hs->slot[hs->cur];
    if (s.filled == 0 && s.in_flight) {  // slot being reused: its previous batch must have landed
        if (int rc = collect(hs, s)) return rc;
    }
    *out_frame = s.h_frames + hs->frame_bytes * (size_t)s.filled;
    hs->acquired = true;
    hs->acquired_n = 1;
    return HVD_OK;
}

int hvd_hasher_commit(hvd_hasher* hs) {
    if (!hs) return hvd::api_fail(HVD_ERR_ARG, "NULL hasher");
    CtxScope scope(hs->ctx);
    if (!hs->acquired) return hvd::api_fail(HVD_ERR_STATE, "hvd_hasher_commit() without hvd_hasher_acquire()");
    if (int rc = hvd::api_bind_device()) return rc;
    hs->acquired = false;
    Slot& s = hs->slot[hs->cur];
    if (++s.filled >= hs->limit) {
        if (int rc = submit(hs, s)) return rc;
        hs->cur = (hs->cur + 1) % kSlots;
    }
    return HVD_OK;
}

/* The same for a RUN of frames (ABI 5): *out_frames is where the next frames belong, *out_n (1 <= *out_n <= want) how many
 * fit there back to back -- what is left of the current batch slot. One call, one pointer, k frames: a decoder that fills
 * 64x64 frames pays one FFI round trip per run instead of two per frame. hvd_hasher_commit_n(n) makes the
 * first n of them count (n <= *out_n; n = 0 is legal and gives the run back). */
int hvd_hasher_acquire_n(hvd_hasher* hs, int64_t want, uint8_t** out_frames, int64_t* out_n) {
    if (!hs || !out_frames || !out_n) return hvd::api_fail(HVD_ERR_ARG, "NULL hasher/out_frames/out_n");
    if (want < 1) return hvd::api_fail(HVD_ERR_ARG, "want must be at least 1");
    if (int rc = hvd_hasher_acquire(hs, out_frames)) return rc;
    const Slot& s = hs->slot[hs->cur];
    *out_n = hs->acquired_n = std::min<int64_t>(want, hs->limit - s.filled);
    return HVD_OK;
}

int hvd_hasher_commit_n(hvd_hasher* hs, int64_t n) {
    if (!hs) return hvd::api_fail(HVD_ERR_ARG, "NULL hasher");
    CtxScope scope(hs->ctx);
    if (!hs->acquired) return hvd::api_fail(HVD_ERR_STATE, "hvd_hasher_commit_n() without hvd_hasher_acquire_n()");
    Slot& s = hs->slot[hs->cur];
    if (n < 0 || n > hs->acquired_n) return hvd::api_fail(HVD_ERR_ARG, "commit of %lld frames, %lld acquired", (long long)n, (long long)hs->acquired_n);
    if (int rc = hvd::api_bind_device()) return rc;
    hs->acquired = false;
    s.filled += n;
    if (s.filled >= hs->limit) {
        if (int rc = submit(hs, s)) return rc;
        hs->cur = (hs->cur + 1) % kSlots;
    }
    return HVD_OK;
}

/* Copies one frame (width*height*channels bytes) into the ring (acquire + memcpy + commit). The caller's
 * buffer is not kept. */
int hvd_hasher_push(hvd_hasher* hs, const uint8_t* frame) {
    if (!hs || !frame) return hvd::api_fail(HVD_ERR_ARG, "NULL hasher/frame");
    uint8_t* dst = nullptr;
    if (int rc = hvd_hasher_acquire(hs, &dst)) return rc;
    {
        NsScope ns(g_ns_copy);
        g_copy_pool.copy(dst, frame, hs->frame_bytes, hs->copy_threads);
    }
    return hvd_hasher_commit(hs);
}

/* Threads that share the host-side copy of one frame in hvd_hasher_push (the caller included): the counterpart of the
 * reference hasher's worker threads. n <= 0: the library default (4). Frames below ~200 KB are copied by the caller. */
int hvd_hasher_set_threads(hvd_hasher* hs, int n) {
    if (!hs) return hvd::api_fail(HVD_ERR_ARG, "NULL hasher");
    hs->copy_threads = n <= 0 ? default_copy_threads() : std::min(n, 1 + hvd::CopyPool::kMaxHelpers);
    return HVD_OK;
}

/* Flushes the partial batch, waits for everything, returns all hashes / qualities in push
 * order (no quality filtering: that is VideoHasher.finish's policy, vpdqpy.py:119). The hasher
 * is empty afterwards and can be reused for the next video. */
int hvd_hasher_finish(hvd_hasher* hs, uint8_t* out_hashes, int32_t* out_quality, int64_t cap, int64_t* out_n) {
    if (!hs || !out_n) return hvd::api_fail(HVD_ERR_ARG, "NULL hasher/out_n");
    CtxScope scope(hs->ctx);
    if (int rc = hvd::api_bind_device()) return rc;
    hs->acquired = false;
    // Slots are submitted in ring order, so the oldest batch still in flight sits in the slot that will be
    // filled next: with nothing staged in the current slot it is the current slot itself (the frame count was
    // a multiple of the batch size), otherwise push() collected it before staging and the oldest is cur+1.
    Slot& c = hs->slot[hs->cur];
    if (c.filled > 0) {
        for (int k = 1; k <= kSlots; ++k) {
            Slot& s = hs->slot[(hs->cur + k) % kSlots];
            if (&s == &c)
                if (int rc = submit(hs, s)) return rc;  // the partial batch is the youngest
            if (int rc = collect(hs, s)) return rc;
        }
    } else {
        for (int k = 0; k < kSlots; ++k)
            if (int rc = collect(hs, hs->slot[(hs->cur + k) % kSlots])) return rc;
    }
    const int64_t n = (int64_t)hs->quality.size();
    *out_n = n;
    if (n > cap) return hvd::api_fail(HVD_ERR_OVERFLOW, "need room for %lld frames, cap %lld", (long long)n, (long long)cap);
    if (n) {
        if (!out_hashes || !out_quality) return hvd::api_fail(HVD_ERR_ARG, "NULL output");
        memcpy(out_hashes, hs->hashes.data(), 32 * (size_t)n);
        memcpy(out_quality, hs->quality.data(), 4 * (size_t)n);
    }
    hs->hashes.clear();
    hs->quality.clear();
    hs->cur = 0;
    hs->limit = first_limit(hs);  // the next video starts on an empty pipeline again
    return HVD_OK;
}

int hvd_hasher_pending(hvd_hasher* hs, int64_t* out_frames) {
    if (!hs || !out_frames) return hvd::api_fail(HVD_ERR_ARG, "NULL");
    int64_t n = (int64_t)hs->quality.size();
    for (Slot& s : hs->slot) n += s.filled + s.in_flight;
    *out_frames = n;
    return HVD_OK;
}

}  // extern "C"
