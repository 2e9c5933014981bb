// hvd_comm.cpp -- the RCCL exchange of the C-ABI (one communicator per context: unique id, init, all-gathers of candidate pairs and
// of raw bytes, destroy / abort) and the grow-only device staging it shares with the scratch pool of a context. Split out of
// hvd_api.cpp in round 6; shared state: hvd_internal.h.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "hvd_internal.h"

using namespace hvdi;

extern "C" {

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
    // the two small words of the video search's agreement step are allocated here, so that nothing can fail between a
    // rank's decision to enter that collective and the collective itself
    // (sized for THIS world: a failed hvd_comm_init used to leave buffers of its own world behind, and a retry with a
    // larger one all-gathered 16 * world bytes into them -- ADVICE r3)
    free_exchange_buffers();
    HIP_TRY(hipMalloc(&g.x_cnt_in, 16));
    HIP_TRY(hipMalloc(&g.x_cnt_all, 16 * (size_t)world));
    {
        ncclResult_t r_ = ncclCommInitRank(&g.comm, world, id, rank);
        if (r_ != ncclSuccess) {
            free_exchange_buffers();
            return fail(HVD_ERR_RCCL, "ncclCommInitRank(world=%d, rank=%d): %s", world, rank, ncclGetErrorString(r_));
        }
    }
    g.comm_ready = true;
    g.rank = rank;
    g.world = world;
    return HVD_OK;
}

}  // extern "C"

namespace hvdi {
void free_exchange_buffers() {
    void** ps[] = {&g.x_cnt_in, &g.x_cnt_all, &g.x_send, &g.x_recv};
    for (void** p : ps) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    g.x_send_cap = g.x_recv_cap = 0;
}
}  // namespace hvdi

extern "C" {

int hvd_comm_destroy(void) {
    std::lock_guard<std::mutex> lk(g_mu);  // same lock as hvd_shutdown, which also tears the communicator down
    if (!g.ready) return HVD_OK;           // hvd_shutdown already destroyed it
    if (int rc = need_ready()) return rc;
    if (g.comm_ready) {
        (void)hipStreamSynchronize(g.stream);
        free_exchange_buffers();
        NCCL_TRY(ncclCommDestroy(g.comm));
        g.comm_ready = false;
    }
    return HVD_OK;
}

int hvd_comm_abort(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g.ready || !g.comm_ready) return HVD_OK;
    if (int rc = need_ready()) return rc;
    g.comm_ready = false;  // whatever ncclCommAbort says, nothing may use this communicator again
    free_exchange_buffers();
    NCCL_TRY(ncclCommAbort(g.comm));
    return HVD_OK;
}

int hvd_comm_allgather_bytes(const void* d_send, void* d_recv, size_t bytes_per_rank) {
    if (int rc = need_ready()) return rc;
    if (g.host_exchange) {  // in-process group without RCCL: through host memory (every context's thread calls this)
        const int W = g.world;
        std::vector<unsigned long long>& mine = g_hx.words[(size_t)g.rank];
        HxGuard hx;
        HX_BARRIER(W);
        mine.resize((bytes_per_rank + 7) / 8);
        if (bytes_per_rank) HIP_TRY(hipMemcpyAsync(mine.data(), d_send, bytes_per_rank, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        HX_BARRIER(W);
        for (int r = 0; r < W && bytes_per_rank; ++r)
            HIP_TRY(hipMemcpyAsync((char*)d_recv + (size_t)r * bytes_per_rank, g_hx.words[(size_t)r].data(), bytes_per_rank,
                                   hipMemcpyHostToDevice, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        HX_BARRIER(W);
        hx.done = true;
        return HVD_OK;
    }
    if (!g.comm_ready) return fail(HVD_ERR_STATE, "hvd_comm_init() has not been called");
    NCCL_TRY(ncclAllGather(d_send, d_recv, bytes_per_rank, ncclUint8, g.comm, g.stream));
    return HVD_OK;
}

}  // extern "C"

namespace hvdi {
int grow(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return HVD_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t want = need < (1u << 16) ? (1u << 16) : need + need / 2;
    HIP_TRY(hipMalloc(p, want));
    *cap = want;
    return HVD_OK;
}

int scratch(Ctx::Scr id, size_t need, void** out) {
    if (int rc = grow(&g.scr[id], &g.scr_cap[id], need ? need : 1)) return rc;
    *out = g.scr[id];
    return HVD_OK;
}
}  // namespace hvdi

extern "C" {

int hvd_comm_allgather_pairs(const void* d_pairs, int64_t count, hvd_pair* out_host, int64_t cap,
                             int64_t* out_total) {
    if (int rc = need_ready()) return rc;
    if (count < 0 || cap < 0 || !out_total) return fail(HVD_ERR_ARG, "bad arguments");
    if (g.host_exchange) {  // in-process group without RCCL: the ranks' records meet in host memory
        const int W = g.world;
        std::vector<unsigned long long>& mine = g_hx.words[(size_t)g.rank];
        HxGuard hx;
        HX_BARRIER(W);
        mine.resize(2 * (size_t)count);
        if (count) HIP_TRY(hipMemcpyAsync(mine.data(), d_pairs, sizeof(hvd_pair) * (size_t)count, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        HX_BARRIER(W);
        size_t total = 0;
        for (int r = 0; r < W; ++r) total += g_hx.words[(size_t)r].size() / 2;
        *out_total = (int64_t)total;
        int rc = HVD_OK;
        if ((int64_t)total > cap) rc = fail(HVD_ERR_OVERFLOW, "need %zu records, cap %lld", total, (long long)cap);
        else if (total && !out_host) rc = fail(HVD_ERR_ARG, "out_host is NULL");
        else {
            size_t o = 0;
            for (int r = 0; r < W; ++r) {
                const size_t m = g_hx.words[(size_t)r].size() / 2;
                if (m) memcpy(out_host + o, g_hx.words[(size_t)r].data(), sizeof(hvd_pair) * m);
                o += m;
            }
        }
        HX_BARRIER(W);
        hx.done = true;  // (an overflow is this rank's own verdict after the exchange: every barrier has been passed)
        return rc;
    }
    if (!g.comm_ready) return fail(HVD_ERR_STATE, "hvd_comm_init() has not been called");
    const int W = g.world;
    if (!g.x_cnt_in || !g.x_cnt_all) return fail(HVD_ERR_STATE, "exchange words missing: hvd_comm_init() allocates them");
    // Round 5: ONE collective in the common case. Every rank sends a fixed slot -- a 16-byte header with its true count, then
    // its first kSlot records -- so that counts and records travel together: one all-gather, one read-back, one
    // synchronisation per step instead of two of each (the step of a strong-scaling run at N = 8 is ~2 ms). Only when some
    // rank holds more than kSlot records does a second all-gather move the remainders, padded to the longest; every rank
    // sees the same headers and takes the same branch.
    static_assert(sizeof(hvd_pair) == 16, "a slot's header (true count, padded) takes the place of one record");
    constexpr size_t kSlot = 1023, kSlotBytes = sizeof(hvd_pair) * (kSlot + 1);
    if (int rc = grow(&g.x_send, &g.x_send_cap, kSlotBytes)) return rc;
    if (int rc = grow(&g.x_recv, &g.x_recv_cap, kSlotBytes * (size_t)W)) return rc;
    const unsigned long long head[2] = {(unsigned long long)count, 0ull};
    const size_t first = std::min<size_t>((size_t)count, kSlot);
    HIP_TRY(hipMemcpyAsync(g.x_send, head, 16, hipMemcpyHostToDevice, g.stream));
    if (first) HIP_TRY(hipMemcpyAsync((char*)g.x_send + sizeof(hvd_pair), d_pairs, sizeof(hvd_pair) * first, hipMemcpyDeviceToDevice, g.stream));
    NCCL_TRY(ncclAllGather(g.x_send, g.x_recv, kSlotBytes, ncclUint8, g.comm, g.stream));
    std::vector<hvd_pair> slots((kSlot + 1) * (size_t)W);
    HIP_TRY(hipMemcpyAsync(slots.data(), g.x_recv, kSlotBytes * (size_t)W, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    std::vector<unsigned long long> counts((size_t)W);
    unsigned long long mx = 0, total = 0;
    for (int r = 0; r < W; ++r) {
        memcpy(&counts[(size_t)r], &slots[(kSlot + 1) * (size_t)r], 8);
        mx = std::max(mx, counts[(size_t)r]);
        total += counts[(size_t)r];
    }
    *out_total = (int64_t)total;
    // (every rank computes the same total and cap is the caller's: ranks pass equal caps in a sharded pass, so all of them
    // leave here together or none does; a second phase is entered by all or none because mx is the same everywhere)
    std::vector<hvd_pair> rest;
    unsigned long long rest_mx = 0;
    if (mx > kSlot) {
        rest_mx = mx - kSlot;
        if (int rc = grow(&g.x_send, &g.x_send_cap, sizeof(hvd_pair) * (size_t)rest_mx)) return rc;
        if (int rc = grow(&g.x_recv, &g.x_recv_cap, sizeof(hvd_pair) * (size_t)rest_mx * (size_t)W)) return rc;
        const size_t mine = (size_t)count > kSlot ? (size_t)count - kSlot : 0;
        if (mine < rest_mx)
            HIP_TRY(hipMemsetAsync((char*)g.x_send + sizeof(hvd_pair) * mine, 0, sizeof(hvd_pair) * (size_t)(rest_mx - mine), g.stream));
        if (mine)
            HIP_TRY(hipMemcpyAsync(g.x_send, (const char*)d_pairs + sizeof(hvd_pair) * kSlot, sizeof(hvd_pair) * mine, hipMemcpyDeviceToDevice, g.stream));
        NCCL_TRY(ncclAllGather(g.x_send, g.x_recv, sizeof(hvd_pair) * (size_t)rest_mx, ncclUint8, g.comm, g.stream));
        rest.resize((size_t)rest_mx * (size_t)W);
        HIP_TRY(hipMemcpyAsync(rest.data(), g.x_recv, sizeof(hvd_pair) * rest.size(), hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
    }
    if ((int64_t)total > cap) return fail(HVD_ERR_OVERFLOW, "need %llu records, cap %lld", total, (long long)cap);
    if (total == 0) return HVD_OK;
    if (!out_host) return fail(HVD_ERR_ARG, "out_host is NULL");
    size_t o = 0;
    for (int r = 0; r < W; ++r) {
        const size_t c = (size_t)counts[(size_t)r], f = std::min(c, kSlot);
        if (f) memcpy(out_host + o, &slots[(kSlot + 1) * (size_t)r + 1], sizeof(hvd_pair) * f);
        o += f;
        if (c > kSlot) {
            memcpy(out_host + o, rest.data() + (size_t)r * (size_t)rest_mx, sizeof(hvd_pair) * (c - kSlot));
            o += c - kSlot;
        }
    }
    return HVD_OK;
}

}  // extern "C"
