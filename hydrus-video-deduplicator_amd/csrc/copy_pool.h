// copy_pool.h -- the helper-thread pool that shares one frame's host memcpy into the pinned ring (hvd_stream.cpp).
// Header-only and free of HIP so that tests/native/copy_pool_tsan.cpp can build it under ThreadSanitizer on a CPU-only box.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <thread>
// (hvd_stream.cpp goes through hipcc, whose device pass also sees this header: CPUID builtins and x86 intrinsics exist in
// the host pass only)
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define HVD_COPY_X86 1
#include <immintrin.h>
#endif

namespace hvd {

// The ring's slots are page-locked, CACHEABLE host memory that the CPU writes once and the DMA engine then reads. A plain
// memcpy store to a line the core does not own first reads it (read for ownership): the write traffic doubles and the
// frame is dragged through the cache on its way to a reader that is not a CPU. copy_stream writes the
// 64-byte-aligned body with NON-TEMPORAL stores instead (AVX-512 or AVX2, chosen by CPUID at run time), plain head and
// tail, one sfence at the end of the slice -- the stores must be globally visible before the slice is reported done, since
// the next reader is the DMA engine. HVD_COPY_NT=0 in the environment (or set_copy_nt(0)) keeps the plain memcpy: the A/B
// switch of tests/test_copy_pool.py and bench.py.
inline std::atomic<int>& copy_nt_flag() {
    static std::atomic<int> f{-1};  // -1: not decided yet
    return f;
}
inline int copy_nt_level() {  // 0 plain memcpy, 2 AVX2 streaming stores, 3 AVX-512 streaming stores
    int v = copy_nt_flag().load(std::memory_order_relaxed);
    if (v >= 0) return v;
    v = 0;
#ifdef HVD_COPY_X86
    const char* e = getenv("HVD_COPY_NT");
    if (!(e && *e == '0')) {
        __builtin_cpu_init();
        v = __builtin_cpu_supports("avx512f") ? 3 : __builtin_cpu_supports("avx2") ? 2 : 0;
    }
#endif
    copy_nt_flag().store(v, std::memory_order_relaxed);
    return v;
}
inline void set_copy_nt(int on) {  // 0: plain memcpy; otherwise: whatever the CPU offers
    copy_nt_flag().store(on ? -1 : 0, std::memory_order_relaxed);
    if (on) {
#ifdef HVD_COPY_X86
        __builtin_cpu_init();
        copy_nt_flag().store(__builtin_cpu_supports("avx512f") ? 3 : __builtin_cpu_supports("avx2") ? 2 : 0, std::memory_order_relaxed);
#else
        copy_nt_flag().store(0, std::memory_order_relaxed);
#endif
    }
}
#ifdef HVD_COPY_X86
__attribute__((target("avx512f"))) inline void stream_body_512(uint8_t* dst, const uint8_t* src, size_t n64) {
    size_t k = 0;
    for (; k + 256 <= n64; k += 256) {  // dst is 64-byte aligned; src need not be
        const __m512i a = _mm512_loadu_si512((const void*)(src + k)), b = _mm512_loadu_si512((const void*)(src + k + 64));
        const __m512i c = _mm512_loadu_si512((const void*)(src + k + 128)), d = _mm512_loadu_si512((const void*)(src + k + 192));
        _mm512_stream_si512((__m512i*)(dst + k), a);
        _mm512_stream_si512((__m512i*)(dst + k + 64), b);
        _mm512_stream_si512((__m512i*)(dst + k + 128), c);
        _mm512_stream_si512((__m512i*)(dst + k + 192), d);
    }
    for (; k < n64; k += 64) _mm512_stream_si512((__m512i*)(dst + k), _mm512_loadu_si512((const void*)(src + k)));
}
__attribute__((target("avx2"))) inline void stream_body_256(uint8_t* dst, const uint8_t* src, size_t n64) {
    for (size_t k = 0; k < n64; k += 64) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(src + k)), b = _mm256_loadu_si256((const __m256i*)(src + k + 32));
        _mm256_stream_si256((__m256i*)(dst + k), a);
        _mm256_stream_si256((__m256i*)(dst + k + 32), b);
    }
}
#endif
inline void copy_stream(uint8_t* dst, const uint8_t* src, size_t n) {
#ifdef HVD_COPY_X86
    const int level = copy_nt_level();
    if (level == 0 || n < 4096) {
        memcpy(dst, src, n);
        return;
    }
    const size_t head = (64 - ((uintptr_t)dst & 63)) & 63;  // (< n: n >= 4096)
    if (head) memcpy(dst, src, head);
    const size_t body = (n - head) & ~(size_t)63;
    if (level == 3)
        stream_body_512(dst + head, src + head, body);
    else
        stream_body_256(dst + head, src + head, body);
    if (n - head - body) memcpy(dst + head + body, src + head + body, n - head - body);
    _mm_sfence();
#else
    memcpy(dst, src, n);
#endif
}

// hash_frame(bytes) is a host memcpy of one frame into the pinned ring; at 512x512 RGB24 one thread moves ~20 GB/s, a
// third of what the PCIe link behind it takes (bench leg videohasher_stream). The reference's VideoHasher owns
// `num_threads` worker threads (vpdqpy/vpdqpy.py:113); here they are what splits that copy: a process-wide pool of
// helper threads, each copying one slice of the frame while the caller copies the first. Helpers spin for a few tens of
// microseconds after a job (frames arrive back to back) and then sleep on a condition variable, so an idle hasher costs
// nothing. Small frames (64x64) are copied by the caller alone.
// Protocol (round 4; the first version published ONE shared job description behind a generation counter, and a helper that
// took no part in job G could read job G+1's fields while still believing it was looking at G -- copy G+1's slice, count
// itself done, then run G+1 again: the caller could return while a slice was still being written):
// every helper has its OWN mailbox. The caller writes a participant's slice into its mailbox and then bumps that helper's
// ticket; a helper reads nothing but its own mailbox, and only after its own ticket moved; the caller does not touch that
// mailbox again before the helper has counted itself done. Helpers that take no part in a job are not involved in it at all.
class CopyPool {
  public:
    static constexpr int kMaxHelpers = 7;
    static constexpr size_t kMinBytesPerThread = 96 << 10;

    ~CopyPool() { stop(); }

    // copy n bytes with up to `threads` threads in total (the caller included)
    void copy(uint8_t* dst, const uint8_t* src, size_t n, int threads) {
        int parts = (int)std::min<size_t>((size_t)std::max(1, threads), n / kMinBytesPerThread);
        if (parts <= 1) {
            if (n >= kMinBytesPerThread) copy_stream(dst, src, n);  // (one thread: still a frame on its way to the DMA engine)
            else memcpy(dst, src, n);
            return;
        }
        std::lock_guard<std::mutex> job_lk(job_mu_);  // one job at a time (two hashers on two threads take turns)
        ensure_helpers(parts - 1);
        parts = std::min(parts, n_helpers_ + 1);
        // (ceiling: with n / parts the last n % parts bytes belonged to nobody whenever n / parts was a multiple of 64 --
        // found by tests/test_copy_pool.py; the reference's 786 432-byte frames divide evenly, other geometries need not)
        const size_t slice = ((n + (size_t)parts - 1) / (size_t)parts + 63) & ~(size_t)63;
        int given = 0;
        for (int id = 1; id < parts; ++id) {
            const size_t off = slice * (size_t)id;
            if (off >= n) break;
            ++given;
        }
        pending_.store(given, std::memory_order_relaxed);
        for (int id = 1; id <= given; ++id) {
            Box& b = box_[id - 1];
            const size_t off = slice * (size_t)id;
            b.src = src + off;
            b.dst = dst + off;
            b.len = std::min(slice, n - off);
            b.ticket.fetch_add(1);  // seq_cst, like the sleepers_ accesses: "publish, then look for sleepers" here against
                                    // "announce sleep, then look at the ticket" in run() is a store-load handshake
        }
        if (sleepers_.load() > 0) {
            std::lock_guard<std::mutex> lk(mu_);
            cv_.notify_all();
        }
        copy_stream(dst, src, std::min(slice, n));
        while (pending_.load(std::memory_order_acquire) > 0) cpu_relax();
    }

    void stop() {
        std::lock_guard<std::mutex> job_lk(job_mu_);  // never under a running copy(): its helpers would vanish
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_.store(true, std::memory_order_release);
            cv_.notify_all();
        }
        for (int i = 0; i < n_helpers_; ++i)
            if (th_[i].joinable()) th_[i].join();
        n_helpers_ = 0;
        stop_.store(false, std::memory_order_release);
    }

  private:
    struct alignas(64) Box {
        std::atomic<uint64_t> ticket{0};
        const uint8_t* src = nullptr;
        uint8_t* dst = nullptr;
        size_t len = 0;
    };
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    void ensure_helpers(int want) {
        want = std::min(want, kMaxHelpers);
        while (n_helpers_ < want) {
            const int idx = n_helpers_;
            const uint64_t seen = box_[idx].ticket.load(std::memory_order_acquire);
            th_[idx] = std::thread([this, idx, seen] { run(idx, seen); });
            ++n_helpers_;
        }
    }
    void run(int idx, uint64_t seen) {
        Box& b = box_[idx];
        for (;;) {
            int spins = 0;
            while (b.ticket.load(std::memory_order_acquire) == seen) {
                if (stop_.load(std::memory_order_acquire)) return;
                if (++spins < 20000) {
                    cpu_relax();
                    continue;
                }
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1);
                cv_.wait(lk, [&] { return stop_.load(std::memory_order_acquire) || b.ticket.load() != seen; });
                sleepers_.fetch_sub(1);
                spins = 0;
            }
            ++seen;  // tickets move by one per job and the caller waits for this helper before the next
            copy_stream(b.dst, b.src, b.len);  // (ends with an sfence: the slice is globally visible before it counts as done)
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }

    std::thread th_[kMaxHelpers];
    Box box_[kMaxHelpers];
    int n_helpers_ = 0;  // guarded by job_mu_
    std::mutex job_mu_, mu_;
    std::condition_variable cv_;
    std::atomic<int> pending_{0}, sleepers_{0};
    std::atomic<bool> stop_{false};
};

}  // namespace hvd
