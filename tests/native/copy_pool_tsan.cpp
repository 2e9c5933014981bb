// ThreadSanitizer stress of the streaming hasher's copy pool (csrc/copy_pool.h), built and run by tests/test_copy_pool.py:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -pthread copy_pool_tsan.cpp -o copy_pool_tsan && ./copy_pool_tsan
// Three callers with different thread counts and frame sizes take turns on ONE pool (the setting of round 3's generation
// race), every copy is compared byte for byte, and in between the pool is stopped and restarted (what
// stream_release_cache does) and left idle long enough for the helpers to go to sleep (the condition-variable path).
// Exit code 0 and no "WARNING: ThreadSanitizer" on stderr = pass.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../hydrus-video-deduplicator_amd/csrc/copy_pool.h"

static hvd::CopyPool pool;
static std::atomic<int> failures{0};

static void hammer(size_t n, int threads, int rounds, unsigned seed) {
    std::vector<uint8_t> src[2] = {std::vector<uint8_t>(n), std::vector<uint8_t>(n)}, dst(n + 64, 0xA5);
    for (auto& v : src)
        for (size_t i = 0; i < n; ++i) v[i] = (uint8_t)((seed = seed * 1664525u + 1013904223u) >> 24);
    for (int r = 0; r < rounds; ++r) {
        const std::vector<uint8_t>& s = src[r & 1];
        pool.copy(dst.data(), s.data(), n, threads);
        if (memcmp(dst.data(), s.data(), n) != 0 || dst[n] != 0xA5) {
            failures.fetch_add(1);
            return;
        }
        if ((r & 63) == 63) std::this_thread::sleep_for(std::chrono::microseconds(300));  // helpers fall asleep now and then
    }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    for (int rep = 0; rep < 3; ++rep) {
        std::thread a(hammer, (size_t)512 * 512 * 3, 2, rounds, 1u + rep);
        std::thread b(hammer, (size_t)512 * 512 * 3, 8, rounds, 2u + rep);
        std::thread c(hammer, (size_t)512 * 512, 3, rounds, 3u + rep);
        std::thread d(hammer, (size_t)1'000'003, 5, rounds / 2, 4u + rep);
        a.join(); b.join(); c.join(); d.join();
        pool.stop();  // helpers are joined; the next copy() starts new ones
    }
    if (failures.load() != 0) {
        fprintf(stderr, "copy mismatch in %d caller(s)\n", failures.load());
        return 1;
    }
    puts("copy_pool_tsan ok");
    return 0;
}
