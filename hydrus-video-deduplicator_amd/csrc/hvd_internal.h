// hvd_internal.h -- what the translation units of the host layer share (round 6: hvd_api.cpp was one 2 200-line unit):
//   hvd_api.cpp     contexts and the device group (init, fan-out, abandon / re-arm), device-resident API, timers, developer keys
//   hvd_search.cpp  host-buffer entry points and the video-level search (K3)
//   hvd_comm.cpp    the RCCL exchange (communicator per context, all-gathers of pairs / bytes)
// Everything here is internal linkage in spirit (namespace hvdi); the C-ABI is include/hvd_mi355x.h alone.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <functional>
#include <mutex>
#include <vector>

#include "host_barrier.h"
#include "hvd_kernels.h"

namespace hvdi {

extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return hvdi::fail(HVD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return hvdi::fail(HVD_ERR_RCCL, "%s: %s", #expr, ncclGetErrorString(r_)); \
    } while (0)

struct Ctx {
    bool ready = false;
    int device = -1;
    int id = 0;               // index in the group (= rank of the in-process sharding)
    bool host_exchange = false;  // group without RCCL (a device listed twice): exchange steps go through host memory
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t mark[8] = {};  // hvd_timer_mark / hvd_timer_between: created on first use
    int v_us[3] = {0, 0, 0};  // last video search on this context, microseconds of host time: local phase (pack, probe, all-pairs
                              // pass, key set), key exchange (agreement, all-gather, merged set), fold (pair map)
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
    hipStream_t m_srv_stream = nullptr;  // the match server's own stream (k_match_server stays resident between calls)
    int32_t m_launch = 0;                // id of the last server launch; hdr[7] == m_launch: that server has left
    std::mutex m_mu;
    // grow-only device scratch of the host-buffer entry points and of the video-level reduction: nothing is
    // allocated or freed per call once the sizes have been seen (h_mu serialises the users)
    enum Scr {
        S_DB, S_IMG, S_GRP, S_PAIRS, S_DB2, S_IMG2, S_GRP2, S_VIDQ, S_VIDT, S_OFF, S_SET, S_SET2, S_PKEYS, S_PCNT, S_LIST,
        S_LISTALL, S_VOUT, S_FRAMES, S_FSCR, S_HASH, S_QUAL, S_COMPACT, S_COUNTERS, S_BITS, S_BITS2, S_BROWS, S_BCOOC, S_BITS_O, S_BITS2_O,
        S_IMG_O, S_IMG2_O, S_N
    };
    void* scr[S_N] = {};
    size_t scr_cap[S_N] = {};
    unsigned long long v_pslots = 0;  // pair map left behind by vmatch_build for vmatch_emit
    int v_exchange_mode = 0;          // hvd_debug_set("vmatch_exchange"): 0 exchange keys iff world > 1, 1 always, 2 never
    int v_fail_rank = 0;              // hvd_debug_set("vmatch_fail_rank"): rank + 1 whose local phase fails (tests the agreement step)
    int v_force_slots_log2 = 0;       // hvd_debug_set("vmatch_slots_log2"): start the tables this small (tests the regrowth)
    int v_variant = 0;                // hvd_debug_set("vmatch_variant"): all-pairs form of the video-level searches, 0 = default (tests, fuzz)
    int v_bit_order = 1;              // hvd_debug_set("vmatch_bit_order"): data-dependent bit order of the video search: 0 never, 1 from 65 536 frames on, 2 always
    int v_bit_order_used = 0;         // the last video search on this context rewrote its hashes in a chosen bit order
    std::recursive_mutex h_mu;
};
constexpr int kMaxCtx = 16;
extern Ctx g_ctx[kMaxCtx];
extern int g_nctx;                 // contexts of the group (0 before hvd_init / hvd_init_devices)
extern bool g_group_rccl;          // the group's contexts hold communicators of one ncclCommInitAll
extern bool g_group_was_rccl;      // ... did when the group was formed (hvd_group_rearm re-creates aborted communicators)
extern thread_local bool t_agreed_exit;  // this context left its last group call through an agreement step, in lock-step with its peers
extern int g_match_server;         // hvd_debug_set "match_server"
extern thread_local int t_ctx;     // the calling thread's current context
#define g (hvdi::g_ctx[hvdi::t_ctx])
extern std::mutex g_mu;

// Rendezvous of the group's worker threads for the exchange steps that have no RCCL underneath: csrc/host_barrier.h (abortable
// generation barrier + a slot of words per rank; TSan-tested on the CPU).
using hvd::HostExchange;
extern HostExchange g_hx;
// every host-memory barrier of a group call: a broken barrier ends the call on this rank too
#define HX_BARRIER(W)                                                                                                      \
    do {                                                                                                                   \
        if (!hvdi::g_hx.barrier(W)) return hvdi::fail(HVD_ERR_RCCL, "group exchange abandoned: another context of the group failed"); \
    } while (0)
struct HxGuard : hvd::HxGuard {  // (host_barrier.h; bound to the group's one exchange)
    HxGuard() : hvd::HxGuard(g_hx) {}
};

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

int need_ready();
inline bool pair_less(const hvd_pair& x, const hvd_pair& y) { return x.i != y.i ? x.i < y.i : x.j < y.j; }

// hvd_api.cpp: the group
int run_on_group(const std::function<int(int)>& fn);  // fn(rank) on every context, one host thread each
int exchange_words(const unsigned long long word[2], std::vector<unsigned long long>& all);  // all-gather of two words per rank inside a group call
void abort_group_comms();

// hvd_comm.cpp: grow-only device staging, scratch pool of a context
void free_exchange_buffers();
int grow(void** p, size_t* cap, size_t need);
int scratch(Ctx::Scr id, size_t need, void** out);
#define SCR(id, bytes, ptr)                                                                  \
    do {                                                                                     \
        if (int rc_ = hvdi::scratch(hvdi::Ctx::id, (bytes), (void**)&(ptr))) return rc_;     \
    } while (0)

}  // namespace hvdi
