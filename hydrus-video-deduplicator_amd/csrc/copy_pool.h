// copy_pool.h -- the helper-thread pool that shares one frame's host memcpy into the pinned ring (hvd_stream.cpp).
// Header-only and free of HIP so that tests/native/copy_pool_tsan.cpp can build it under ThreadSanitizer on a CPU-only box.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>

namespace hvd {

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
            memcpy(dst, src, n);
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
        memcpy(dst, src, std::min(slice, n));
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
            memcpy(b.dst, b.src, b.len);
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
