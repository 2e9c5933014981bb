// ThreadSanitizer stress of the device group's host-memory rendezvous (csrc/host_barrier.h), built and run by
// tests/test_copy_pool.py::test_host_barrier_under_thread_sanitizer:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -pthread host_barrier_tsan.cpp -o host_barrier_tsan && ./host_barrier_tsan
// The shape of hvd_api.cpp's exchange blocks (run_on_group -> exchange_words / all-gather through g_hx.words): W rank threads per
// "group call"; every rank writes its slot between two barriers and reads everybody's after the second. Calls alternate
// between clean ones (every rank's sum must be right), calls in which one rank leaves early through an HxGuard (its peers must
// come out of their barrier with `false`, nobody may hang, nobody may read a slot that is being written) and calls aborted from
// OUTSIDE (hvd_group_abort from another thread); rearm() between calls, as run_on_group does.
// Exit code 0 and no "WARNING: ThreadSanitizer" on stderr = pass.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../../hydrus-video-deduplicator_amd/csrc/host_barrier.h"

static hvd::HostExchange hx;
static std::atomic<int> failures{0}, abandoned{0}, completed{0};

// one rank's part of one group call; fail_at: the rank that leaves before the exchange (-1: nobody), rounds of exchange per call
static bool rank_body(int rank, int W, int call, int fail_at, int rounds) {
    for (int r = 0; r < rounds; ++r) {
        hvd::HxGuard guard(hx);
        if (rank == fail_at && r == rounds / 2) return false;  // (the guard breaks the barrier on the way out)
        if (!hx.barrier(W)) return false;                       // everybody is done with the previous round's slots
        hx.words[(size_t)rank].assign({(unsigned long long)(call * 1000 + r), (unsigned long long)rank});
        if (!hx.barrier(W)) return false;
        unsigned long long sum = 0;
        for (int k = 0; k < W; ++k) {
            if (hx.words[(size_t)k].size() != 2 || hx.words[(size_t)k][0] != (unsigned long long)(call * 1000 + r)) failures.fetch_add(1);
            sum += hx.words[(size_t)k][1];
        }
        if (sum != (unsigned long long)(W * (W - 1) / 2)) failures.fetch_add(1);
        guard.done = true;
    }
    return true;
}

int main(int argc, char** argv) {
    const int calls = argc > 1 ? atoi(argv[1]) : 300;
    for (int W : {2, 3, 8}) {
        hx.words.assign((size_t)W, {});
        for (int call = 0; call < calls; ++call) {
            hx.rearm();
            const int mode = call % 4;  // 0, 1: clean; 2: a rank leaves early; 3: aborted from outside
            const int fail_at = mode == 2 ? call % W : -1;
            std::atomic<int> ok{0};
            std::vector<std::thread> th;
            for (int rank = 0; rank < W; ++rank)
                th.emplace_back([&, rank] {
                    if (rank_body(rank, W, call, fail_at, 4)) ok.fetch_add(1);
                });
            std::thread outsider;
            if (mode == 3) outsider = std::thread([&] {
                std::this_thread::sleep_for(std::chrono::microseconds(50 + 37 * (call % 5)));
                hx.abort();
            });
            for (auto& t : th) t.join();  // (a hang here is the failure this test exists for: the runner's timeout catches it)
            if (outsider.joinable()) outsider.join();
            if (mode < 2) {
                if (ok.load() != W) failures.fetch_add(1);
                completed.fetch_add(1);
            } else if (mode == 2) {
                if (ok.load() != 0 && ok.load() != W - 1) {}  // peers may finish the rounds before the failure; never the failing rank
                if (ok.load() == W) failures.fetch_add(1);
                abandoned.fetch_add(1);
            } else {
                abandoned.fetch_add(1);  // (an outside abort may arrive after the call has finished: any outcome but a hang is fine)
            }
        }
    }
    if (failures.load()) {
        printf("host_barrier_tsan FAILED: %d\n", failures.load());
        return 1;
    }
    printf("host_barrier_tsan ok (%d clean calls, %d abandoned)\n", completed.load(), abandoned.load());
    return 0;
}
