// host_barrier.h -- rendezvous of the device group's worker threads for the exchange steps that have no RCCL underneath
// (hvd_api.cpp: a group that lists one device twice, or whose communicators could not be created). HIP-free and header-only,
// so that the same code is stressed under ThreadSanitizer on a CPU (tests/native/host_barrier_tsan.cpp), like copy_pool.h.
//
// A plain generation barrier plus a slot of words per rank. The barrier can be ABORTED: a rank that leaves a group call with an
// error -- a HIP failure between two barriers, a context that is not ready -- breaks it, so that its peers come out of their
// wait with `false` and return an error instead of waiting for ever with the group mutex held. rearm() puts it back to work
// once everybody has left (run_on_group, hvd_group_rearm).
#pragma once
#include <condition_variable>
#include <mutex>
#include <vector>

namespace hvd {

struct HostExchange {
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long gen = 0;
    bool broken = false;
    std::vector<std::vector<unsigned long long>> words;  // one vector per rank: written by its owner between two barriers
    bool barrier(int n) {
        std::unique_lock<std::mutex> lk(mu);
        if (broken) return false;
        const unsigned long long my = gen;
        if (++arrived == n) {
            arrived = 0;
            ++gen;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return gen != my || broken; });
        }
        return gen != my;  // (completed: true even if a rank that left through it has broken it since)
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
    void rearm() {
        std::lock_guard<std::mutex> lk(mu);
        broken = false;
        arrived = 0;
    }
};

// Declared at the top of every host-memory exchange block: whoever leaves the block early (an error return, a broken barrier)
// breaks the barrier for its peers on the way out; the normal exit -- after the block's last barrier -- disarms it.
struct HxGuard {
    HostExchange& hx;
    bool done = false;
    explicit HxGuard(HostExchange& h) : hx(h) {}
    ~HxGuard() {
        if (!done) hx.abort();
    }
};

}  // namespace hvd
