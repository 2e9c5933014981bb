// hvd_api.cpp -- host side of the C-ABI declared in include/hvd_mi355x.h.
//
// A process drives one MI355X (hvd_init(device)) or a GROUP of them (hvd_init_devices / HVD_DEVICES): one context per
// listed device -- stream, timer events, grow-only scratch pool, select/context words of the all-pairs kernel, RCCL
// communicator -- the "ranks" of the tile-cyclic sharding inside one process (the reference is one process:
// entrypoint.py:235 -> dedup.py:213). Every entry point works on the calling thread's CURRENT context (context 0 unless
// hvd_set_context says otherwise); the host-buffer entry points fan out over all contexts of the group by themselves, one
// host thread per context, so that whatever binds this library uses every configured GPU without a launcher. No CPU
// fallback exists anywhere in this file: every compute entry point needs an initialised device and fails with
// HVD_ERR_STATE / HVD_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "host_barrier.h"
#include "hvd_kernels.h"
#include "../../include/hvd_mi355x_bench.h"

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

// Multi-process GPU work on hosts whose driver only supports dmabuf IPC needs HSA_ENABLE_IPC_MODE_LEGACY=0 (without it RCCL's
// hipIpcGetMemHandle fails with "invalid argument"). The variable is read when the HSA runtime starts, i.e. at the first HIP
// call: set it when the library is loaded, never over a value the user chose.
__attribute__((constructor)) void hvd_default_ipc_mode() { setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", /*overwrite=*/0); }

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
Ctx g_ctx[kMaxCtx];
int g_nctx = 0;                 // contexts of the group (0 before hvd_init / hvd_init_devices)
bool g_group_rccl = false;      // the group's contexts hold communicators of one ncclCommInitAll
bool g_group_was_rccl = false;  // ... did when the group was formed (hvd_group_rearm re-creates aborted communicators)
thread_local bool t_agreed_exit = false;  // this context left its last group call through an agreement step, in lock-step with its peers
int g_match_server = 1;         // hvd_debug_set "match_server": hvd_match_two's small operands go to a resident workgroup (1) or to one launch per call (0)
constexpr unsigned long long kMatchServerIdleUs = 300;  // the server leaves after this long without a call ...
constexpr unsigned long long kMatchServerLifeUs = 2000;  // ... and after this long in any case (another thread's hipFree / device-wide wait gets its turn)
thread_local int t_ctx = 0;     // the calling thread's current context
#define g (g_ctx[t_ctx])
std::mutex g_mu;

// Rendezvous of the group's worker threads for the exchange steps that have no RCCL underneath: csrc/host_barrier.h (abortable
// generation barrier + a slot of words per rank; TSan-tested on the CPU).
using hvd::HostExchange;
HostExchange g_hx;
// every host-memory barrier of a group call: a broken barrier ends the call on this rank too
#define HX_BARRIER(W)                                                                                              \
    do {                                                                                                            \
        if (!g_hx.barrier(W)) return fail(HVD_ERR_RCCL, "group exchange abandoned: another context of the group failed"); \
    } while (0)

struct HxGuard : hvd::HxGuard {  // (host_barrier.h; bound to the group's one exchange)
    HxGuard() : hvd::HxGuard(g_hx) {}
};

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
    if (t_ctx >= g_nctx) t_ctx = 0;  // (a thread that selected a context of an earlier, larger group)
    if (!g.ready) return fail(HVD_ERR_STATE, "hvd_init() has not been called (no CPU fallback exists)");
    // HIP's current device is per host thread; the reference calls this path from the main thread or
    // from a QThread worker (gui/gui.py:195-237), so every entry re-asserts the bound device.
    hipError_t e = hipSetDevice(g.device);
    if (e != hipSuccess) return fail(HVD_ERR_HIP, "hipSetDevice(%d): %s", g.device, hipGetErrorString(e));
    return HVD_OK;
}

bool pair_less(const hvd_pair& x, const hvd_pair& y) { return x.i != y.i ? x.i < y.i : x.j < y.j; }

}  // namespace

extern "C" {  // (defined further down inside the extern "C" block: g++ insists that declaration and definition agree)
static void free_exchange_buffers();
static int grow(void** p, size_t* cap, size_t need);
}

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
int api_bind_device() { return need_ready(); }
int api_context() { return t_ctx; }
void api_set_context(int idx) { t_ctx = idx; }
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
    hvd::pdq_dct_table_copy(out);
    return HVD_OK;
}

int hvd_dct_matrix_libm(float* out) {
    if (!out) return fail(HVD_ERR_ARG, "out is NULL");
    fill_dct(out);
    return HVD_OK;
}

}  // extern "C" (context management helpers have C++ linkage)

namespace {

// Bring context `idx` up on HIP device `device` (the caller holds g_mu and has checked the device list).
int init_context(int idx, int device) {
    const int saved = t_ctx;
    t_ctx = idx;
    struct Restore {
        int v;
        ~Restore() { t_ctx = v; }
    } restore{saved};
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(HVD_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    // The DCT matrix is a constant of the algorithm: the table compiled into the kernels (csrc/dct_table.inc, generated
    // once by scripts/gen_dct_table.py) is authoritative for the literal AND the operand forms of the hash kernel, so the
    // hashes do not depend on this host's libm. tests/ compare it with hvd_dct_matrix_libm() and with the oracle.
    hvd::pdq_dct_table_copy(g.h_dct);
    hipError_t e = hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&g.ev0);
    if (e == hipSuccess) e = hipEventCreate(&g.ev1);
    if (e == hipSuccess) e = hipMalloc((void**)&g.d_dct, sizeof g.h_dct);
    if (e == hipSuccess) e = hipMemcpy(g.d_dct, g.h_dct, sizeof g.h_dct, hipMemcpyHostToDevice);
    if (e != hipSuccess) {  // a failed init leaves nothing behind, so that a retry does not leak
        if (g.d_dct) (void)hipFree(g.d_dct);
        if (g.ev1) (void)hipEventDestroy(g.ev1);
        if (g.ev0) (void)hipEventDestroy(g.ev0);
        if (g.stream) (void)hipStreamDestroy(g.stream);
        g.d_dct = nullptr;
        g.ev0 = g.ev1 = nullptr;
        g.stream = nullptr;
        return fail(HVD_ERR_HIP, "hvd_init(%d): %s", device, hipGetErrorString(e));
    }
    g.device = device;
    g.id = idx;
    g.ready = true;
    return HVD_OK;
}

void shutdown_context(int idx) {
    const int saved = t_ctx;
    t_ctx = idx;
    if (g.ready) {
        (void)hipSetDevice(g.device);
        (void)hipStreamSynchronize(g.stream);
        if (g.comm_ready) {
            free_exchange_buffers();
            (void)ncclCommDestroy(g.comm);
            g.comm_ready = false;
        }
        (void)hipFree(g.d_dct);
        (void)hipEventDestroy(g.ev0);
        (void)hipEventDestroy(g.ev1);
        for (hipEvent_t m : g.mark)
            if (m) (void)hipEventDestroy(m);
        (void)hipStreamDestroy(g.stream);
        for (void** p : {&g.m_a, &g.m_b, &g.m_f, &g.m_o})
            if (*p) (void)hipFree(*p);
        for (void* p : g.scr)
            if (p) (void)hipFree(p);
        if (g.m_srv_stream) {  // (a resident match server leaves by itself within its idle limit)
            (void)hipStreamSynchronize(g.m_srv_stream);
            (void)hipStreamDestroy(g.m_srv_stream);
        }
        if (g.m_pin) (void)hipHostFree(g.m_pin);
        g.~Ctx();
        new (&g) Ctx();
    }
    t_ctx = saved;
}

int parse_dct_mode(int* out) {
    *out = hvd::g_pdq_dct_mode;
    if (const char* m = getenv("HVD_PDQ_DCT_MODE")) {  // same switch as hvd_set_pdq_dct_mode(); checked before any resource exists
        if (!strcmp(m, "fma") || !strcmp(m, "1")) *out = HVD_DCT_FMA;
        else if (!strcmp(m, "strict") || !strcmp(m, "0") || !*m) *out = HVD_DCT_STRICT;
        else return fail(HVD_ERR_ARG, "HVD_PDQ_DCT_MODE=%s: expected strict or fma", m);
    }
    return HVD_OK;
}

}  // namespace

namespace {
// The group's communicators are created, aborted and re-created from different threads (hvd_init_devices, hvd_group_abort,
// hvd_group_rearm, a failing context of run_on_group): one mutex around every transition, and `comm_ready` is claimed under it
// before ncclCommAbort runs, so that no communicator is aborted twice.
std::mutex g_comm_mu;
void abort_group_comms() {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (int k = 0; k < g_nctx; ++k)
        if (g_ctx[k].comm_ready) {
            g_ctx[k].comm_ready = false;
            (void)ncclCommAbort(g_ctx[k].comm);
        }
}

// The exchange steps of the sharded searches: RCCL all-gathers between the group's devices (ncclCommInitAll: one process,
// one communicator per device). RCCL refuses a device that is listed twice -- such a group (a test configuration: two
// contexts on one GPU) exchanges through host memory instead, and so does a group whose communicators cannot be created;
// hvd_group_exchange() says which. Also the recovery path of hvd_group_rearm (communicators aborted after a failure).
void form_group_exchange(bool distinct) {
    const int n_devices = g_nctx;
    g_group_rccl = false;
    if (n_devices <= 1) return;
    std::lock_guard<std::mutex> lk(g_comm_mu);
    bool rccl = distinct && !getenv("HVD_GROUP_NO_RCCL");
    if (rccl) {
        ncclComm_t comms[kMaxCtx];
        int devices[kMaxCtx];
        for (int i = 0; i < n_devices; ++i) devices[i] = g_ctx[i].device;
        ncclResult_t r = ncclCommInitAll(comms, n_devices, devices);
        const int saved = t_ctx;
        if (r == ncclSuccess) {
            for (int i = 0; i < n_devices; ++i) {  // every communicator has an owner first (so that a failure below aborts all)
                g_ctx[i].comm = comms[i];
                g_ctx[i].comm_ready = true;
            }
            for (int i = 0; i < n_devices && rccl; ++i) {
                t_ctx = i;
                (void)hipSetDevice(g.device);
                g.rank = i;
                g.world = n_devices;
                if (!g.x_cnt_in && hipMalloc(&g.x_cnt_in, 16) != hipSuccess) rccl = false;
                if (!g.x_cnt_all && hipMalloc(&g.x_cnt_all, 16 * (size_t)n_devices) != hipSuccess) rccl = false;
            }
        } else {
            rccl = false;
        }
        if (!rccl) {
            for (int i = 0; i < n_devices; ++i)
                if (g_ctx[i].comm_ready) {
                    t_ctx = i;
                    (void)hipSetDevice(g.device);
                    free_exchange_buffers();
                    (void)ncclCommAbort(g.comm);
                    g.comm_ready = false;
                }
            (void)hipGetLastError();
        }
        t_ctx = saved;
    }
    g_group_rccl = rccl;
    for (int i = 0; i < n_devices; ++i) {
        g_ctx[i].host_exchange = !rccl;
        g_ctx[i].rank = i;
        g_ctx[i].world = n_devices;
    }
}

bool group_devices_distinct() {
    for (int i = 0; i < g_nctx; ++i)
        for (int k = 0; k < i; ++k)
            if (g_ctx[k].device == g_ctx[i].device) return false;
    return true;
}

// Put a group whose exchange was abandoned back to work: the host barrier is re-armed; an RCCL group whose communicators were
// aborted (hvd_group_abort, a hard failure inside a sharded call) gets new ones. Nobody may be inside a group call.
int rearm_group() {
    g_hx.rearm();
    if (g_nctx <= 1 || !g_group_was_rccl) return HVD_OK;
    bool whole = true;
    for (int k = 0; k < g_nctx; ++k) whole = whole && g_ctx[k].comm_ready;
    if (whole) return HVD_OK;
    abort_group_comms();  // (a half-aborted set: finish the job, then form the group again)
    form_group_exchange(group_devices_distinct());
    if (!g_group_rccl) return fail(HVD_ERR_RCCL, "the group's RCCL communicators could not be re-created (exchanging through host memory now)");
    return HVD_OK;
}
}  // namespace

extern "C" {

int hvd_init_devices(const int* devices, int n_devices) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!devices || n_devices < 1 || n_devices > kMaxCtx)
        return fail(HVD_ERR_ARG, "hvd_init_devices: need 1..%d devices", kMaxCtx);
    if (g_nctx > 0) {
        bool same = g_nctx == n_devices;
        for (int i = 0; same && i < n_devices; ++i) same = g_ctx[i].device == devices[i];
        if (same) return rearm_group();  // (idempotent; a group whose exchange was aborted is formed again: include/hvd_mi355x.h)
        return fail(HVD_ERR_STATE, "already bound to %d device(s) starting with device %d; hvd_shutdown() first", g_nctx,
                    g_ctx[0].device);
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(HVD_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    }
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= n) return fail(HVD_ERR_ARG, "device %d out of range [0,%d)", devices[i], n);
        for (int k = 0; k < i; ++k) distinct = distinct && devices[k] != devices[i];
    }
    int dct_mode = 0;
    if (int rc = parse_dct_mode(&dct_mode)) return rc;
    for (int i = 0; i < n_devices; ++i)
        if (int rc = init_context(i, devices[i])) {
            for (int k = 0; k < i; ++k) shutdown_context(k);
            return rc;
        }
    hvd::g_pdq_dct_mode = dct_mode;
    g_nctx = n_devices;
    g_group_rccl = false;
    g_hx.words.assign((size_t)n_devices, {});
    g_hx.rearm();
    form_group_exchange(distinct);
    g_group_was_rccl = g_group_rccl;
    (void)hipSetDevice(g_ctx[0].device);
    return HVD_OK;
}

int hvd_init(int device) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_nctx > 0) {
            if (g_nctx == 1 && g_ctx[0].device == device) return HVD_OK;
            if (g_nctx > 1 && g_ctx[0].device == device) return HVD_OK;  // a group whose first context is this device
            return fail(HVD_ERR_STATE, "already bound to device %d (hvd_shutdown() first, or list the devices: hvd_init_devices)",
                        g_ctx[0].device);
        }
    }
    // HVD_DEVICES=0,1,2,3 turns the one-device initialisation every binding performs into a group: the drop-in surfaces
    // (search, VpTreeManager facade, SQLite adapter, pipeline) then shard over all listed GPUs inside this process.
    if (const char* env = getenv("HVD_DEVICES")) {
        int devs[kMaxCtx], n = 0;
        const char* p = env;
        while (*p && n < kMaxCtx) {
            char* end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) return fail(HVD_ERR_ARG, "HVD_DEVICES=%s: expected a comma-separated list of device numbers", env);
            devs[n++] = (int)v;
            p = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') return fail(HVD_ERR_ARG, "HVD_DEVICES=%s: expected a comma-separated list of device numbers", env);
        }
        if (n > 1 || (n == 1 && devs[0] != device)) {
            if (n >= 1 && devs[0] != device)
                return fail(HVD_ERR_ARG, "hvd_init(%d) with HVD_DEVICES=%s: the list must start with the device asked for", device, env);
            return hvd_init_devices(devs, n);
        }
    }
    return hvd_init_devices(&device, 1);
}

int hvd_context_count(int* out_n) {
    if (!out_n) return fail(HVD_ERR_ARG, "out_n is NULL");
    *out_n = g_nctx;
    return HVD_OK;
}

int hvd_set_context(int index) {
    if (index < 0 || index >= (g_nctx > 0 ? g_nctx : 1)) return fail(HVD_ERR_ARG, "context %d out of range [0,%d)", index, g_nctx);
    t_ctx = index;
    if (g.ready) HIP_TRY(hipSetDevice(g.device));
    return HVD_OK;
}

int hvd_get_context(void) { return t_ctx; }

int hvd_group_exchange(void) { return g_nctx <= 1 ? 0 : g_group_rccl ? 1 : 2; }

int hvd_group_abort(void) {
    g_hx.abort();
    if (g_group_rccl) abort_group_comms();
    return HVD_OK;
}

int hvd_group_rearm(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_nctx <= 0) return fail(HVD_ERR_STATE, "hvd_group_rearm before hvd_init / hvd_init_devices");
    return rearm_group();
}

int hvd_runtime_info(char* buf, size_t len) {
    if (!buf || len == 0) return fail(HVD_ERR_ARG, "buf is NULL");
    std::string o = "{";
    auto add = [&](const char* fmt, ...) {
        char tmp[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(tmp, sizeof tmp, fmt, ap);
        va_end(ap);
        o += tmp;
    };
    int rt = 0, drv = 0, nccl = 0, ndev = 0;
    (void)hipRuntimeGetVersion(&rt);
    (void)hipDriverGetVersion(&drv);
    (void)ncclGetVersion(&nccl);
    if (hipGetDeviceCount(&ndev) != hipSuccess) {
        (void)hipGetLastError();
        ndev = 0;
    }
    Dl_info di;
    const char* rccl_path = dladdr((void*)&ncclGetVersion, &di) && di.dli_fname ? di.dli_fname : "?";
    const char* hip_path = dladdr((void*)&hipRuntimeGetVersion, &di) && di.dli_fname ? di.dli_fname : "?";
    add("\"abi\": %d, \"hip_runtime_version\": %d, \"hip_driver_version\": %d, \"rccl_version\": %d, \"rccl_built_against\": %d, "
        "\"librccl_path\": \"%s\", \"libamdhip64_path\": \"%s\", \"visible_devices\": %d, \"devices\": [",
        HVD_ABI_VERSION, rt, drv, nccl, NCCL_VERSION_CODE, rccl_path, hip_path, ndev);
    for (int d = 0; d < ndev && d < 16; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        char bus[32] = "?";
        (void)hipDeviceGetPCIBusId(bus, sizeof bus, d);
        add("%s{\"index\": %d, \"name\": \"%s\", \"arch\": \"%s\", \"pci\": \"%s\", \"cus\": %d, \"mem_gib\": %.1f, \"clock_mhz\": %d}",
            d ? ", " : "", d, p.name, p.gcnArchName, bus, p.multiProcessorCount, (double)p.totalGlobalMem / (1 << 30), p.clockRate / 1000);
    }
    o += "], \"group\": [";
    for (int k = 0; k < g_nctx; ++k) add("%s%d", k ? ", " : "", g_ctx[k].device);
    add("], \"group_exchange\": \"%s\", \"peers\": [", g_nctx <= 1 ? "none" : g_group_rccl ? "rccl" : "host");
    // peer matrix of the group's devices (all visible ones if there is no group): access + link type + hops
    std::vector<int> devs;
    for (int k = 0; k < g_nctx; ++k) devs.push_back(g_ctx[k].device);
    if (devs.size() <= 1) {
        devs.clear();
        for (int d = 0; d < ndev && d < 16; ++d) devs.push_back(d);
    }
    bool first = true;
    for (size_t a = 0; a < devs.size(); ++a)
        for (size_t b = 0; b < devs.size(); ++b) {
            if (devs[a] == devs[b]) continue;
            int can = 0;
            uint32_t link = 0, hops = 0;
            if (hipDeviceCanAccessPeer(&can, devs[a], devs[b]) != hipSuccess) (void)hipGetLastError();
            if (hipExtGetLinkTypeAndHopCount(devs[a], devs[b], &link, &hops) != hipSuccess) {
                (void)hipGetLastError();
                link = 0xFFFFFFFFu;
            }
            // hsa_amd_link_info_type_t: 0 HyperTransport, 1 QPI, 2 PCIe, 3 InfiniBand, 4 xGMI
            const char* lname = link == 4 ? "xgmi" : link == 2 ? "pcie" : link == 0xFFFFFFFFu ? "?" : "other";
            add("%s{\"from\": %d, \"to\": %d, \"access\": %d, \"link\": \"%s\", \"link_type\": %d, \"hops\": %u}", first ? "" : ", ",
                devs[a], devs[b], can, lname, (int)link, hops);
            first = false;
        }
    o += "]}";
    snprintf(buf, len, "%s", o.c_str());
    return o.size() < len ? HVD_OK : fail(HVD_ERR_OVERFLOW, "runtime info needs %zu bytes", o.size() + 1);
}

int hvd_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_nctx == 0) return HVD_OK;
    for (int i = 0; i < g_nctx; ++i) shutdown_context(i);
    hvd::stream_release_cache();
    hvd::mfma_release();
    hvd::pdq_release();
    g_nctx = 0;
    g_group_rccl = false;
    t_ctx = 0;
    return HVD_OK;
}

/* ------------------------------------------------------------ device API -- */

int hvd_dev_malloc(void** out_ptr, size_t bytes) {
    if (int rc = need_ready()) return rc;
    if (!out_ptr) return fail(HVD_ERR_ARG, "out_ptr is NULL");
    HIP_TRY(hipMalloc(out_ptr, bytes ? bytes : 1));
    return HVD_OK;
}

int hvd_host_malloc(void** out_ptr, size_t bytes) {
    if (int rc = need_ready()) return rc;
    if (!out_ptr) return fail(HVD_ERR_ARG, "out_ptr is NULL");
    HIP_TRY(hipHostMalloc(out_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return HVD_OK;
}

int hvd_host_free(void* h_ptr) {
    if (int rc = need_ready()) return rc;
    if (h_ptr) HIP_TRY(hipHostFree(h_ptr));
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

int hvd_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes) {
    if (int rc = need_ready()) return rc;
    if (bytes) HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, g.stream));
    return HVD_OK;
}

int hvd_dev_sync(void) {
    if (int rc = need_ready()) return rc;
    HIP_TRY(hipStreamSynchronize(g.stream));
    return HVD_OK;
}

int hvd_device_synchronize(void) {
    if (int rc = need_ready()) return rc;
    HIP_TRY(hipDeviceSynchronize());
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

int hvd_timer_mark(int slot) {
    if (int rc = need_ready()) return rc;
    if (slot < 0 || slot >= 8) return fail(HVD_ERR_ARG, "timer slot %d: 0..7", slot);
    if (!g.mark[slot]) HIP_TRY(hipEventCreate(&g.mark[slot]));
    HIP_TRY(hipEventRecord(g.mark[slot], g.stream));
    return HVD_OK;
}

int hvd_timer_between(int slot_a, int slot_b, float* out_ms) {
    if (int rc = need_ready()) return rc;
    if (!out_ms || slot_a < 0 || slot_a >= 8 || slot_b < 0 || slot_b >= 8) return fail(HVD_ERR_ARG, "timer slots 0..7, out_ms not NULL");
    if (!g.mark[slot_a] || !g.mark[slot_b]) return fail(HVD_ERR_STATE, "hvd_timer_between: a slot was never marked");
    HIP_TRY(hipEventSynchronize(g.mark[slot_b]));
    HIP_TRY(hipEventElapsedTime(out_ms, g.mark[slot_a], g.mark[slot_b]));
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
        if (value < 0 || value > 3) return fail(HVD_ERR_ARG, "pdq_dct_from_lds: 0 SGPR operands, 1 LDS, 2 literals, 3 by batch size");
        hvd::g_pdq_dct_from_lds = value;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_hash_grid") == 0) {
        if (value < 0) return fail(HVD_ERR_ARG, "pdq_hash_grid must not be negative (0 = default)");
        hvd::g_pdq_hash_grid = value;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_hash_prefetch") == 0) {  // 64x64 gray hash kernel, static launches: next frame fetched one frame ahead
        hvd::g_pdq_hash_prefetch = value != 0;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_luma_lut") == 0) {
        hvd::g_pdq_luma_lut = value;
        return HVD_OK;
    }
    if (strcmp(key, "mfma_col_chunk_max") == 0) {
        if (value < 256 || value % 128) return fail(HVD_ERR_ARG, "mfma_col_chunk_max must be a multiple of 128, >= 256");
        hvd::g_mfma_col_chunk_max = (uint32_t)value;
        return HVD_OK;
    }
    if (strcmp(key, "mfma_auto_mid") == 0) {  // form the auto variant runs on data with common false survivors
        if (value != 0 && value != 18) return fail(HVD_ERR_ARG, "mfma_auto_mid: 18 (panel-mark queue form) or 0 (none: fetch or register form only)");
        hvd::g_mfma_auto_mid = (uint32_t)value;
        return HVD_OK;
    }
    if (strcmp(key, "mfma_force_sel") == 0) {  // which 128 bits the first stage sees: -1 the probe's choice, 0 bits 0..127, 1 bits 128..255, 2 bits 0..63 + 192..255
        if (value < -1 || value > 2) return fail(HVD_ERR_ARG, "mfma_force_sel: -1 (the probe chooses) or 0 | 1 | 2");
        hvd::g_mfma_force_sel = value;
        return HVD_OK;
    }
    if (strcmp(key, "mfma_queue_packed") == 0) {  // 0: the pair-queue form settles its candidates from the FP4 images only
        hvd::g_mfma_queue_packed = value != 0;
        return HVD_OK;
    }
    if (strcmp(key, "mfma_auto_mid_max_x100") == 0) {  // ... up to this many survivors per 1024-pair tile (x 0.01)
        if (value < 0) return fail(HVD_ERR_ARG, "mfma_auto_mid_max_x100 must not be negative");
        hvd::g_mfma_auto_mid_max_x100 = (uint32_t)value;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_down512_wave") == 0) {
        if (value < 0 || value > 2) return fail(HVD_ERR_ARG, "pdq_down512_wave: 0 never, 1 by batch size, 2 always");
        hvd::g_pdq_down512_wave = value;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_down512_strip") == 0) {
        if (value != 0 && value != 32 && value != 64) return fail(HVD_ERR_ARG, "pdq_down512_strip: 0 by batch size, 32, 64");
        hvd::g_pdq_down512_strip = value;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_down512_wave_grid") == 0) {
        if (value < 0) return fail(HVD_ERR_ARG, "pdq_down512_wave_grid must not be negative (0 = default)");
        hvd::g_pdq_down512_wave_grid = value;
        return HVD_OK;
    }
    if (strcmp(key, "vmatch_slots_log2") == 0) {
        if (value != 0 && (value < 4 || value > 30)) return fail(HVD_ERR_ARG, "vmatch_slots_log2: 0 (automatic) or 4..30");
        for (int k = 0; k < g_nctx; ++k) g_ctx[k].v_force_slots_log2 = value;  // (every context of the group: ADVICE r4)
        return HVD_OK;
    }
    if (strcmp(key, "vmatch_variant") == 0) {  // tests / scripts/gpu_fuzz_k3.py: the video-level searches through one explicit form
        if (value != 0 && value != 8 && value != 9 && value != 12 && value != 13 && value != 18)
            return fail(HVD_ERR_ARG, "vmatch_variant: 0 (default) or an MFMA form: 8, 9, 12, 13 (auto), 18");
        // every context of the group: forms differ in their tile height, so ranks on different forms would walk different
        // (rb + cb) % world partitions -- tiles skipped or compared twice
        for (int k = 0; k < g_nctx; ++k) g_ctx[k].v_variant = value;
        return HVD_OK;
    }
    if (strcmp(key, "vmatch_bit_order") == 0) {  // data-dependent bit order of the video search: 0 never, 1 from 65 536 frames on (default), 2 always
        if (value < 0 || value > 2) return fail(HVD_ERR_ARG, "vmatch_bit_order: 0 | 1 | 2");
        for (int k = 0; k < std::max(1, g_nctx); ++k) g_ctx[k].v_bit_order = value;
        return HVD_OK;
    }
#ifndef HVD_NO_BENCH_SYMBOLS
    if (strcmp(key, "vmatch_fail_rank") == 0) {  // tests only (include/hvd_mi355x_bench.h): rank (value - 1) fails before the key exchange; 0 = off
        for (int k = 0; k < g_nctx; ++k) g_ctx[k].v_fail_rank = value;
        return HVD_OK;
    }
#endif
    if (strcmp(key, "vmatch_exchange") == 0) {
        if (value < 0 || value > 2) return fail(HVD_ERR_ARG, "vmatch_exchange: 0 iff world > 1, 1 always, 2 never (partial results)");
        for (int k = 0; k < g_nctx; ++k) g_ctx[k].v_exchange_mode = value;
        return HVD_OK;
    }
    if (strcmp(key, "pdq_fused_down512") == 0) {
        hvd::g_pdq_fused_down512 = value != 0;
        return HVD_OK;
    }
    if (strcmp(key, "match_server") == 0) {  // hvd_match_two, small operands: 1 resident match server (default), 0 one launch per call
        g_match_server = value != 0;
        return HVD_OK;
    }
    if (strcmp(key, "copy_nt") == 0) {  // hvd_hasher_push: non-temporal stores into the pinned ring (1, default where the CPU has them) or plain memcpy (0)
        hvd::stream_set_copy_nt(value);
        return HVD_OK;
    }
    if (strcmp(key, "mfma_clock_reset") == 0) {  // telemetry: clear this context's clock accumulators (in stream order)
        if (int rc = need_ready()) return rc;
        HIP_TRY(hvd::mfma_clock_reset(t_ctx, g.stream));
        return HVD_OK;
    }
    return fail(HVD_ERR_ARG, "unknown debug key %s", key);
}

int hvd_debug_get(const char* key, int* out_value) {
    if (!key || !out_value) return fail(HVD_ERR_ARG, "NULL argument");
    if (strcmp(key, "match_server_ticks") == 0) {  // -DHVD_MATCH_SERVER_TIMING builds: 10 ns ticks the server spent on the last request
        const size_t small = hvd::match_two_small_limit();
        *out_value = g.m_pin ? reinterpret_cast<volatile int32_t*>(g.m_pin + small)[8] : 0;
        return HVD_OK;
    }
    if (strcmp(key, "copy_nt") == 0) {  // (host only: no device needed)
        *out_value = hvd::stream_copy_nt_level();
        return HVD_OK;
    }
    {
        const char* hk[3] = {"hasher_us_copy", "hasher_us_submit", "hasher_us_wait"};
        for (int k = 0; k < 3; ++k)
            if (strcmp(key, hk[k]) == 0) {  // host microseconds of the streaming feed since the last read (process-wide; reading clears)
                *out_value = (int)std::min<long long>(hvd::stream_take_ns(k) / 1000, 0x7FFFFFFF);
                return HVD_OK;
            }
    }
    if (int rc = need_ready()) return rc;
    // what the probe of the last auto-variant launch saw and chose: form id, survivors over bits 0..127 / 128..255,
    // 1 if the first stage ran on bits 128..255
    // (word 4 is the probe's ticket; word 5 the survivors over bits 0..63 + 192..255, round 5)
    const char* keys[6] = {"mfma_auto_form", "mfma_probe_survivors", "mfma_probe_survivors_hi", "mfma_auto_half", "", "mfma_probe_survivors_mix"};
    for (int k = 0; k < 6; ++k)
        if (keys[k][0] && strcmp(key, keys[k]) == 0) {
            uint32_t* sel = nullptr;
            HIP_TRY(hvd::mfma_select_buffer(t_ctx, &sel));
            uint32_t v[6] = {0, 0, 0, 0, 0, 0};
            HIP_TRY(hipMemcpyAsync(v, sel, 24, hipMemcpyDeviceToHost, g.stream));
            HIP_TRY(hipStreamSynchronize(g.stream));
            *out_value = (int)v[k];
            return HVD_OK;
        }
    if (strcmp(key, "vmatch_bit_order_used") == 0) {
        *out_value = g.v_bit_order_used;
        return HVD_OK;
    }
    {
        const char* vk[3] = {"vmatch_us_local", "vmatch_us_exchange", "vmatch_us_fold"};
        for (int k = 0; k < 3; ++k)
            if (strcmp(key, vk[k]) == 0) {
                *out_value = g.v_us[k];
                return HVD_OK;
            }
    }
    if (strcmp(key, "mfma_pass_khz") == 0 || strcmp(key, "mfma_clock_samples") == 0) {
        // telemetry (round 6): the shader clock the FP4-MFMA all-pairs passes of this context actually ran at since the last
        // "mfma_clock_reset" -- sampled workgroups' s_memtime cycles over their s_memrealtime ticks (k_allpairs_mfma) x the
        // tick rate the runtime reports (hipDeviceAttributeWallClockRate, kHz); waits for the stream. 0: nothing sampled.
        unsigned long long v[4] = {0, 0, 0, 0};
        HIP_TRY(hvd::mfma_clock_read(t_ctx, g.stream, v));
        if (key[5] == 'c') {
            *out_value = (int)std::min<unsigned long long>(v[2], 0x7FFFFFFFull);
            return HVD_OK;
        }
        int dev = 0, wall_khz = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev));
        *out_value = v[1] ? (int)((double)v[0] / (double)v[1] * (double)wall_khz + 0.5) : 0;
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
    if (variant != 8 && variant != 9 && variant != 12 && variant != 13 && variant != 18)
        return fail(HVD_ERR_ARG, "unknown FP4-MFMA variant %d (8, 9, 12, 18, or 13 = chosen by the probe)", variant);
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
    a.ctx_id = t_ctx;
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
    if (variant != 0 && variant != 1) return fail(HVD_ERR_ARG, "unknown popcount variant %d (0 | 1)", variant);
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
    a.ctx_id = t_ctx;
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
    // packed hashes for the pair-queue form, derived from the images into the pool (this entry is handed images only)
    std::lock_guard<std::recursive_mutex> lk(g.h_mu);
    void *d_bits_t = nullptr, *d_bits_q = nullptr;
    if (int rc = grow(&g.scr[Ctx::S_BITS], &g.scr_cap[Ctx::S_BITS], 32 * (size_t)nt)) return rc;
    if (int rc = grow(&g.scr[Ctx::S_BITS2], &g.scr_cap[Ctx::S_BITS2], 32 * (size_t)nq)) return rc;
    d_bits_t = g.scr[Ctx::S_BITS];
    d_bits_q = g.scr[Ctx::S_BITS2];
    HIP_TRY(hvd::launch_pack_fp4(d_img_t, (uint32_t)nt, d_bits_t, g.stream));
    HIP_TRY(hvd::launch_pack_fp4(d_img_q, (uint32_t)nq, d_bits_q, g.stream));
    hvd::AllPairsArgs a;
    a.d_db = d_bits_t;
    a.d_db_q = d_bits_q;
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
    a.ctx_id = t_ctx;
    hipError_t e = hvd::launch_cross_mfma(a, d_img_q, (uint32_t)nq, d_img_t, (const int32_t*)d_group_t, g.stream);
    if (e != hipSuccess) return fail(HVD_ERR_HIP, "launch_cross_mfma: %s", hipGetErrorString(e));
    return HVD_OK;
}

/* ------------------------------------------------- host-buffer entry points -- */

}  // extern "C"

namespace {
// Run fn(rank) on every context of the group, one host thread per context (the caller's thread takes context 0; a group of
// one runs inline). Returns the first failure's code with its message in the calling thread's error buffer. Only ONE fan-out
// runs at a time (the host-memory exchange has one set of slots, and two concurrent fan-outs would only take turns on the
// devices anyway).
std::mutex g_group_mu;
int run_on_group(const std::function<int(int)>& fn) {
    const int n = g_nctx;
    if (n <= 1) return fn(0);
    std::lock_guard<std::mutex> lk(g_group_mu);
    g_hx.rearm();
    std::vector<int> rc((size_t)n, HVD_OK);
    std::vector<std::string> msg((size_t)n);
    std::once_flag rccl_abort;
    auto body = [&](int i) {
        t_ctx = i;
        t_agreed_exit = false;
        rc[(size_t)i] = need_ready();
        if (rc[(size_t)i] == HVD_OK) rc[(size_t)i] = fn(i);
        if (rc[(size_t)i] != HVD_OK) {
            msg[(size_t)i] = g_err;
            // A rank that leaves with an error may leave peers waiting for it in an exchange step. Host-memory
            // group: break the barrier. RCCL group: a HARD failure (HIP / RCCL / state -- not the overflow and argument
            // verdicts, which every rank reaches together after the exchange) aborts the group's communicators, which
            // releases a peer blocked in a collective on the device; the group then has no exchange until it is
            // initialised again (every later sharded call fails loudly instead of hanging).
            g_hx.abort();
            // Not when the failure was AGREED: a rank that fails its local phase -- an out-of-memory, the injected
            // test failure -- reports it through the agreement all-gather, every rank sees it there and all of them return
            // together: nobody is stranded, and the communicators stay usable for the next call.
            const int code = rc[(size_t)i];
            if (g_group_rccl && !t_agreed_exit && (code == HVD_ERR_HIP || code == HVD_ERR_RCCL || code == HVD_ERR_STATE))
                std::call_once(rccl_abort, [&] { abort_group_comms(); });
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n; ++i) th.emplace_back(body, i);
    const int saved = t_ctx;
    body(0);
    t_ctx = saved;
    for (std::thread& t : th) t.join();
    (void)need_ready();
    g_hx.rearm();  // (everybody has left: a barrier broken by this call's own failure -- or by an overflow verdict, which every rank
                   // reaches together -- must not fail the next exchange of callers that drive the contexts from their own threads)
    for (int i = 0; i < n; ++i)
        if (rc[(size_t)i] != HVD_OK) {
            snprintf(g_err, sizeof g_err, "%s", msg[(size_t)i].c_str());
            return rc[(size_t)i];
        }
    return HVD_OK;
}

// all-gather of two words per rank inside a sharded group call (RCCL or host memory); every context's thread calls it
int exchange_words(const unsigned long long word[2], std::vector<unsigned long long>& all) {
    const int W = g.world;
    all.assign(2 * (size_t)W, 0ull);
    if (g.host_exchange) {
        HxGuard hx;
        HX_BARRIER(W);
        g_hx.words[(size_t)g.rank].assign(word, word + 2);
        HX_BARRIER(W);
        for (int r = 0; r < W; ++r) {
            all[2 * (size_t)r] = g_hx.words[(size_t)r][0];
            all[2 * (size_t)r + 1] = g_hx.words[(size_t)r][1];
        }
        hx.done = true;
        return HVD_OK;
    }
    if (!g.comm_ready) return fail(HVD_ERR_STATE, "no communicator on context %d", g.id);
    HIP_TRY(hipMemcpyAsync(g.x_cnt_in, word, 16, hipMemcpyHostToDevice, g.stream));
    NCCL_TRY(ncclAllGather(g.x_cnt_in, g.x_cnt_all, 2, ncclUint64, g.comm, g.stream));
    HIP_TRY(hipMemcpyAsync(all.data(), g.x_cnt_all, 16 * (size_t)W, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return HVD_OK;
}
}  // namespace

extern "C" {

static int scratch(Ctx::Scr id, size_t need, void** out) {
    if (int rc = grow(&g.scr[id], &g.scr_cap[id], need ? need : 1)) return rc;
    *out = g.scr[id];
    return HVD_OK;
}
#define SCR(id, bytes, ptr) \
    do {                    \
        if (int rc_ = scratch(Ctx::id, (bytes), (void**)&(ptr))) return rc_; \
    } while (0)

static int hash_frames_host(const uint8_t* frames, int64_t n, int h, int w, int channels, uint8_t* out_hashes,
                            int32_t* out_quality) {
    if (int rc = need_ready()) return rc;
    if (n < 0 || h < 64 || w < 64) return fail(HVD_ERR_ARG, "bad frame geometry n=%lld h=%d w=%d", (long long)n, h, w);
    if (n == 0) return HVD_OK;
    if (!frames || !out_hashes || !out_quality) return fail(HVD_ERR_ARG, "NULL buffer");
    std::lock_guard<std::recursive_mutex> lk(g.h_mu);
    const size_t frame_bytes = (size_t)h * w * channels;
    // Batches bound the staging footprint (<= ~1 GiB of frames per batch).
    int64_t batch = (int64_t)((1ull << 30) / frame_bytes);
    if (batch < 1) batch = 1;
    if (batch > n) batch = n;
    const bool need_scratch = !(h == 64 && w == 64 && channels == 1);
    void *d_in = nullptr, *d_scr = nullptr, *d_h = nullptr, *d_q = nullptr;
    SCR(S_FRAMES, frame_bytes * batch, d_in);
    if (need_scratch) {
        size_t sb = 0;
        if (int rc = hvd_pdq_scratch_bytes(batch, h, w, channels, &sb)) return rc;
        SCR(S_FSCR, sb, d_scr);
    }
    SCR(S_HASH, 32 * (size_t)batch, d_h);
    SCR(S_QUAL, 4 * (size_t)batch, d_q);
    for (int64_t f0 = 0; f0 < n; f0 += batch) {
        const int64_t m = std::min(batch, n - f0);
        HIP_TRY(hipMemcpyAsync(d_in, frames + frame_bytes * f0, frame_bytes * m, hipMemcpyHostToDevice, g.stream));
        if (int rc = hvd_dev_pdq_hash_frames(d_in, m, h, w, channels, need_scratch ? d_scr : nullptr, d_h, d_q)) return rc;
        HIP_TRY(hipMemcpyAsync(out_hashes + 32 * f0, d_h, 32 * (size_t)m, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipMemcpyAsync(out_quality + f0, d_q, 4 * (size_t)m, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
    }
    return HVD_OK;
}

// frames are independent: a group hashes contiguous ranges of them, one per context, no exchange
static int hash_frames_group(const uint8_t* frames, int64_t n, int h, int w, int channels, uint8_t* out_hashes,
                             int32_t* out_quality) {
    const int W = g_nctx;
    if (W <= 1 || n < 4 * (int64_t)W || !frames || !out_hashes || !out_quality || h < 64 || w < 64)
        return hash_frames_host(frames, n, h, w, channels, out_hashes, out_quality);
    const size_t frame_bytes = (size_t)h * w * channels;
    return run_on_group([&](int r) -> int {
        const int64_t per = (n + W - 1) / W, lo = std::min<int64_t>(n, per * r), hi = std::min<int64_t>(n, lo + per);
        return hash_frames_host(frames + frame_bytes * (size_t)lo, hi - lo, h, w, channels, out_hashes + 32 * lo, out_quality + lo);
    });
}

int hvd_pdq_hash_frames_gray_u8(const uint8_t* frames, int64_t n, int h, int w, uint8_t* out_hashes,
                                int32_t* out_quality) {
    return hash_frames_group(frames, n, h, w, 1, out_hashes, out_quality);
}

int hvd_pdq_hash_frames_rgb24_u8(const uint8_t* frames, int64_t n, int h, int w, uint8_t* out_hashes,
                                 int32_t* out_quality) {
    return hash_frames_group(frames, n, h, w, 3, out_hashes, out_quality);
}

// Runs the default all-pairs kernel (FP4-MFMA form) on a host DB -- this context's share of the tiles (rank of world) --
// and fetches up to `cap` unordered records: its own when world == 1, every rank's after the group's exchange otherwise
// (RCCL all-gather of counts then padded records between the devices, or host memory where the group has no RCCL).
// *out_count = the true number of records over all ranks. Device buffers come from the grow-only pool (caller holds h_mu).
static int allpairs_host_raw(const uint8_t* db, int64_t n, const int32_t* group, int max_dist,
                             std::vector<hvd_pair>& recs, int64_t cap, int64_t* out_count, int rank = 0, int world = 1) {
    void* d_pairs = nullptr;
    unsigned long long cnt = 0;
    // Everything up to the exchange runs inside `local`: at world > 1 its result code rides along with the count, so that a
    // rank that fails on its own does not leave the others waiting in the exchange (as in the video search, vmatch_build).
    auto local = [&]() -> int {
        void *d_db = nullptr, *d_img = nullptr, *d_grp = nullptr;
        unsigned long long* d_cnt = nullptr;
        SCR(S_DB, 32 * (size_t)n, d_db);
        HIP_TRY(hipMemcpyAsync(d_db, db, 32 * (size_t)n, hipMemcpyHostToDevice, g.stream));
        size_t img_bytes = 0;
        if (int rc = hvd_fp4_image_bytes(n, &img_bytes)) return rc;
        SCR(S_IMG, img_bytes, d_img);
        if (int rc = hvd_dev_expand_fp4(d_db, n, d_img)) return rc;
        if (group) {
            SCR(S_GRP, 4 * (size_t)n, d_grp);
            HIP_TRY(hipMemcpyAsync(d_grp, group, 4 * (size_t)n, hipMemcpyHostToDevice, g.stream));
        }
        SCR(S_PAIRS, sizeof(hvd_pair) * (size_t)cap, d_pairs);
        SCR(S_COUNTERS, 64, d_cnt);
        HIP_TRY(hipMemsetAsync(d_cnt, 0, 8, g.stream));
        if (int rc = hvd_dev_allpairs_hamming256_mfma(d_db, d_img, n, group ? d_grp : nullptr, max_dist, rank, world, d_pairs,
                                                      cap, d_cnt, HVD_DEFAULT_VARIANT))
            return rc;
        HIP_TRY(hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        return HVD_OK;
    };
    const int local_rc = local();
    const size_t mine = (size_t)std::min<unsigned long long>(cnt, (unsigned long long)cap);
    if (world == 1) {
        if (local_rc) return local_rc;
        *out_count = (int64_t)cnt;
        recs.resize(mine);
        if (mine) {
            HIP_TRY(hipMemcpyAsync(recs.data(), d_pairs, sizeof(hvd_pair) * mine, hipMemcpyDeviceToHost, g.stream));
            HIP_TRY(hipStreamSynchronize(g.stream));
        }
        return HVD_OK;
    }
    // the true counts first (a rank whose own buffer overflowed must not truncate the total), then the records
    const unsigned long long word[2] = {local_rc ? 0ull : cnt, local_rc ? 1ull : 0ull};
    std::vector<unsigned long long> all;
    if (int rc = exchange_words(word, all)) return local_rc ? local_rc : rc;
    unsigned long long total = 0;
    for (int r = 0; r < world; ++r) {
        if (all[2 * (size_t)r + 1]) return local_rc ? local_rc : fail(HVD_ERR_RCCL, "all-pairs search abandoned: rank %d failed", r);
        total += all[2 * (size_t)r];
    }
    *out_count = (int64_t)total;
    if (total > (unsigned long long)cap) {  // every rank sees the same total: all of them skip the record exchange
        recs.clear();
        return HVD_OK;
    }
    recs.resize((size_t)total);
    int64_t got = 0;
    if (int rc = hvd_comm_allgather_pairs(d_pairs, (int64_t)mine, recs.data(), (int64_t)total, &got)) return rc;
    if (got != (int64_t)total) return fail(HVD_ERR_RCCL, "candidate exchange returned %lld records, expected %llu", (long long)got, total);
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
    const int W = (g_nctx > 1 && max_dist < 128 && n >= 4096) ? g_nctx : 1;  // small DBs: one device (launch-bound anyway)
    int64_t total = 0;
    if (W == 1) {
        std::lock_guard<std::recursive_mutex> lk(g.h_mu);
        if (int rc = allpairs_host_raw(db, n, group, max_dist, recs, cap, &total)) return rc;
    } else {
        // DB replicated on every device of the group, tile (rb, cb) -> context (rb + cb) % W, candidates exchanged
        int rc = run_on_group([&](int r) -> int {
            std::lock_guard<std::recursive_mutex> lk(g.h_mu);
            std::vector<hvd_pair> mine;
            int64_t t = 0;
            if (int rc_ = allpairs_host_raw(db, n, group, max_dist, mine, cap, &t, r, W)) return rc_;
            if (r == 0) {
                recs.swap(mine);
                total = t;
            }
            return HVD_OK;
        });
        if (rc) return rc;
    }
    *out_count = total;
    if (*out_count > cap)
        return fail(HVD_ERR_OVERFLOW, "pair buffer too small: need %lld records, cap %lld", (long long)*out_count,
                    (long long)cap);
    std::sort(recs.begin(), recs.end(), pair_less);
    if (!recs.empty()) memcpy(out, recs.data(), sizeof(hvd_pair) * recs.size());
    return HVD_OK;
}

// Frame-level hits -> per video pair (a = video of the row frame, b = video of the column frame):
// q_hits = distinct row frames, t_hits = distinct column frames. Output sorted by (a, b).
// Only the popcount route (max_dist >= 128, never used by the reference) still reduces on the host.
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
            HIP_TRY(hipHostMalloc((void**)&g.m_pin, small + 64, hipHostMallocCoherent));  // (fine-grained: a RUNNING kernel sees the host's stores)
            memset(g.m_pin + small, 0, 64);
        }
        uint8_t* pb = g.m_pin + 32 * (size_t)na;
        volatile int32_t* ph = reinterpret_cast<volatile int32_t*>(g.m_pin + small);
        memcpy(g.m_pin, a, 32 * (size_t)na);
        memcpy(pb, b, 32 * (size_t)nb);
        const int32_t seq = ++g.m_seq == 0 ? ++g.m_seq : g.m_seq;
        if (g_match_server) {
            // Round 5: post the request to the resident match server (k_match_server) and poll for the answer -- no launch and
            // no synchronisation per call while calls come back to back (the VP-tree's pattern); the server is (re)started
            // when it has left (idle for kMatchServerIdleUs) or has never run.
            const uint32_t seq21 = (uint32_t)seq & 0x1FFFFFu;
            auto start_server = [&]() -> int {
                if (!g.m_srv_stream) HIP_TRY(hipStreamCreateWithFlags(&g.m_srv_stream, hipStreamNonBlocking));
                g.m_launch = g.m_launch == 0x7FFFFFFF ? 1 : g.m_launch + 1;
                HIP_TRY(hvd::launch_match_server((const uint32_t*)g.m_pin, (int32_t*)(g.m_pin + small), (seq21 - 1u) & 0x1FFFFFu,
                                                 g.m_launch, 100ull * kMatchServerIdleUs, 100ull * kMatchServerLifeUs, g.m_srv_stream));
                return HVD_OK;
            };
            // ONE 64-bit word carries the whole request: a poll that sees the new sequence number has everything
            const unsigned long long word = ((unsigned long long)seq21 << 43) | ((unsigned long long)(uint32_t)max_dist << 32) |
                                            ((unsigned long long)(uint32_t)na << 16) | (unsigned long long)(uint32_t)nb;
            __atomic_store_n(reinterpret_cast<volatile unsigned long long*>(ph + 4), word, __ATOMIC_RELEASE);
            if (g.m_launch == 0 || __atomic_load_n(&ph[3], __ATOMIC_ACQUIRE) == g.m_launch)
                if (int rc = start_server()) return rc;
            bool seen = false;
            for (int attempt = 0; attempt < 3 && !seen; ++attempt) {
                for (long spin = 0; spin < 40000000; ++spin) {
                    if ((uint32_t)__atomic_load_n(&ph[2], __ATOMIC_ACQUIRE) == seq21) {
                        seen = true;
                        break;
                    }
                    // the server may have left between our look at hdr[3] and its last poll: start another, it finds the request
                    if ((spin & 1023) == 1023 && __atomic_load_n(&ph[3], __ATOMIC_ACQUIRE) == g.m_launch) break;
                }
                if (!seen) {
                    if (__atomic_load_n(&ph[3], __ATOMIC_ACQUIRE) != g.m_launch) break;  // still running and silent: give up below
                    if (int rc = start_server()) return rc;
                }
            }
            if (!seen) {
                HIP_TRY(hipStreamSynchronize(g.m_srv_stream));
                if ((uint32_t)__atomic_load_n(&ph[2], __ATOMIC_ACQUIRE) != seq21) return fail(HVD_ERR_HIP, "match server did not answer");
            }
            *q_hits = ph[0];
            *t_hits = ph[1];
            return HVD_OK;
        }
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

/* ------------------------------------------ video-level search on the device (K3) -- */

}  // extern "C"

namespace {

unsigned long long pow2_at_least(unsigned long long x) {
    unsigned long long p = 1;
    while (p < x) p <<= 1;
    return p;
}

struct VmArgs {
    const void* d_img_q;  // == d_img_t in the symmetric form
    uint32_t nq;
    const void* d_img_t;
    uint32_t nt;
    bool rect;
    const int32_t *d_vid_q, *d_vid_t;    // video index of every frame (== each other in the symmetric form)
    const int32_t *d_excl_q, *d_excl_t;  // rect only: frames with equal values are not compared (nullable)
    int max_dist;                        // [0,127]
    int rank, world;
    int pre_rc = 0;                      // a failure of this rank BEFORE the search (upload): reported through the agreement step
};

int read_counters(unsigned long long* d_counters, unsigned long long out[4]) {
    HIP_TRY(hipMemcpyAsync(out, d_counters, 32, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return HVD_OK;
}

// Which 128 bits should the first stage see? (k_hamming_mfma.hip: "data-dependent bit order".) From the co-occurrence counts of a
// strided sample of the packed hashes: Pearson correlation of every pair of bits, then 128 times drop the bit whose summed
// |correlation| with the bits still in the set is largest (a constant bit goes first). perm = the 128 kept bits in ascending
// order, then the dropped ones: bit k of a rewritten hash is bit perm[k] of the original. Deterministic in the data, so every rank
// of a sharded search -- the library is replicated -- arrives at the same order. *changed = false: too few hashes, keep the order.
int choose_bit_order(const void* d_bits, uint32_t n, bool always, uint8_t perm[256], bool* changed) {
    *changed = false;
    for (int k = 0; k < 256; ++k) perm[k] = (uint8_t)k;
    const uint32_t sample = std::min<uint32_t>(n, 16384u) & ~63u;
    if (sample < (always ? 64u : 4096u)) return HVD_OK;
    const uint32_t words = sample / 64u, stride = n / sample;
    void *d_rows = nullptr, *d_cooc = nullptr;
    SCR(S_BROWS, 8 * 256 * (size_t)words, d_rows);
    SCR(S_BCOOC, 4 * 256 * 256, d_cooc);
    HIP_TRY(hvd::launch_bit_cooc(d_bits, stride, words, d_rows, d_cooc, g.stream));
    std::vector<uint32_t> cooc(256 * 256);
    HIP_TRY(hipMemcpyAsync(cooc.data(), d_cooc, 4 * cooc.size(), hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    const double N = (double)sample;
    std::vector<double> pr(256), sd(256), a(256 * 256, 0.0), load(256, 0.0);
    for (int i = 0; i < 256; ++i) {
        pr[i] = cooc[(size_t)i * 257] / N;
        sd[i] = std::sqrt(std::max(0.0, pr[i] * (1.0 - pr[i])));
    }
    for (int i = 0; i < 256; ++i)
        for (int j = 0; j < 256; ++j) {
            if (i == j) continue;
            const double r = (sd[i] < 1e-6 || sd[j] < 1e-6) ? 1.0 : (cooc[(size_t)i * 256 + j] / N - pr[i] * pr[j]) / (sd[i] * sd[j]);
            a[(size_t)i * 256 + j] = std::fabs(r);
            load[i] += std::fabs(r);
        }
    bool in[256];
    for (int i = 0; i < 256; ++i) in[i] = true;
    for (int step = 0; step < 128; ++step) {
        int worst = -1;
        for (int i = 0; i < 256; ++i)
            if (in[i] && (worst < 0 || load[i] >= load[worst])) worst = i;  // (ties: the higher bit goes)
        in[worst] = false;
        for (int j = 0; j < 256; ++j) load[j] -= a[(size_t)worst * 256 + j];
    }
    int k = 0;
    for (int i = 0; i < 256; ++i)
        if (in[i]) perm[k++] = (uint8_t)i;
    for (int i = 0; i < 256; ++i)
        if (!in[i]) perm[k++] = (uint8_t)i;
    for (int i = 0; i < 256; ++i) *changed = *changed || perm[i] != i;
    return HVD_OK;
}

// All-pairs pass in video mode -> set of (frame, video) keys -> [key exchange between ranks] -> pair map with
// the vPDQ counters, left in the pool for vmatch_emit. Overflowing tables are rebuilt larger and only the
// step that overflowed is repeated; the inputs never move.
int vmatch_build(const VmArgs& v) {
    const bool exchange = g.v_exchange_mode == 1 || (g.v_exchange_mode == 0 && v.world > 1);
    if (exchange && ((!g.comm_ready && !g.host_exchange) || g.world != v.world || g.rank != v.rank))
        return fail(HVD_ERR_STATE, "rank %d of %d needs hvd_comm_init() with the same rank/world first", v.rank, v.world);
    // world > 1: a rank that fails on its own (out of memory while a table regrows, a launch error) must not leave its
    // peers blocked in the all-gathers below. Everything up to the exchange runs inside `local`, whose result code rides
    // along with the key count in the first all-gather: every rank learns of a failure anywhere and all of them return.
    unsigned long long* d_counters = nullptr;
    unsigned long long slots = 0;
    unsigned long long* d_set = nullptr;
    unsigned long long c[4] = {0, 0, 0, 0};
    auto local = [&]() -> int {
    if (v.pre_rc) return v.pre_rc;
    SCR(S_COUNTERS, 64, d_counters);
    // the pair-queue form of the all-pairs kernel settles its candidates on PACKED hashes; this entry is handed images only
    void *d_bits_t = nullptr, *d_bits_q = nullptr;
    const void *img_t = v.d_img_t, *img_q = v.d_img_q;
    SCR(S_BITS, 32 * (size_t)v.nt, d_bits_t);
    HIP_TRY(hvd::launch_pack_fp4(v.d_img_t, v.nt, d_bits_t, g.stream));
    if (v.rect) {
        SCR(S_BITS2, 32 * (size_t)v.nq, d_bits_q);
        HIP_TRY(hvd::launch_pack_fp4(v.d_img_q, v.nq, d_bits_q, g.stream));
    }
    // Round 5: the search runs on hashes rewritten in a bit order chosen from the library itself (choose_bit_order): the first
    // stage then sees the 128 least entangled bits. Library scratch only -- the caller's image is left as it is -- and the
    // same order for rows and columns, so every distance, and with it every record, is what it was.
    g.v_bit_order_used = 0;
    if (g.v_bit_order == 2 || (g.v_bit_order == 1 && v.nt >= 65536u)) {
        uint8_t perm[256];
        bool changed = false;
        if (int rc = choose_bit_order(d_bits_t, v.nt, g.v_bit_order == 2, perm, &changed)) return rc;
        if (changed) {
            size_t img_bytes = 0;
            void *d_bo = nullptr, *d_io = nullptr;
            if (int rc = hvd_fp4_image_bytes((int64_t)v.nt, &img_bytes)) return rc;
            SCR(S_BITS_O, 32 * (size_t)v.nt, d_bo);
            SCR(S_IMG_O, img_bytes, d_io);
            HIP_TRY(hvd::launch_reorder_bits(d_bits_t, v.nt, perm, d_bo, d_io, g.stream));
            d_bits_t = d_bo;
            img_t = d_io;
            if (v.rect) {
                if (int rc = hvd_fp4_image_bytes((int64_t)v.nq, &img_bytes)) return rc;
                SCR(S_BITS2_O, 32 * (size_t)v.nq, d_bo);
                SCR(S_IMG2_O, img_bytes, d_io);
                HIP_TRY(hvd::launch_reorder_bits(d_bits_q, v.nq, perm, d_bo, d_io, g.stream));
                d_bits_q = d_bo;
                img_q = d_io;
            } else {
                img_q = img_t;
            }
            g.v_bit_order_used = 1;
        }
    }
    const unsigned long long frames = (unsigned long long)v.nt + (v.rect ? v.nq : 0u);
    slots = pow2_at_least(std::max<unsigned long long>(1ull << 16, 4ull * frames));
    if (g.v_force_slots_log2) slots = 1ull << g.v_force_slots_log2;
#ifndef HVD_NO_BENCH_SYMBOLS
    if (g.v_fail_rank == v.rank + 1) return fail(HVD_ERR_HIP, "injected failure on rank %d (hvd_debug_set vmatch_fail_rank)", v.rank);
#endif
    for (;;) {
        SCR(S_SET, 8 * slots, d_set);
        HIP_TRY(hipMemsetAsync(d_set, 0xFF, 8 * slots, g.stream));
        HIP_TRY(hipMemsetAsync(d_counters, 0, 32, g.stream));
        hvd::AllPairsArgs a;
        a.d_db = d_bits_t;
        a.d_db_q = d_bits_q;
        a.n = v.nt;
        a.sync_decide = true;  // (this call waits for its result anyway)
        a.d_group = v.rect ? v.d_excl_q : v.d_vid_q;  // symmetric: frames of one video never match each other
        a.max_dist = (uint32_t)v.max_dist;
        a.rank = (uint32_t)v.rank;
        a.world = (uint32_t)v.world;
        a.d_pairs = nullptr;
        a.cap = 0;
        a.d_count = d_counters + 3;
        a.variant = g.v_variant ? g.v_variant : HVD_DEFAULT_VARIANT;
        a.col_chunk = 0;
        a.ctx_id = t_ctx;
        a.sink = hvd::VideoSink{d_set, slots - 1, d_counters, v.d_vid_q, v.d_vid_t};
        hipError_t e = v.rect ? hvd::launch_cross_mfma(a, img_q, v.nq, img_t, v.d_excl_t, g.stream)
                              : hvd::launch_allpairs_mfma(a, img_t, g.stream);
        if (e != hipSuccess) return fail(HVD_ERR_HIP, "video-level all-pairs launch: %s", hipGetErrorString(e));
        if (int rc = read_counters(d_counters, c)) return rc;
        if (c[0] == 0) break;
        slots *= 4;  // some insert ran out of probes: larger table, same pass again
    }
    return HVD_OK;
    };
    const auto t_begin = std::chrono::steady_clock::now();
    auto us_since = [](std::chrono::steady_clock::time_point t0) {
        return (int)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    };
    const int local_rc = local();  // (ends in read_counters: the stream is drained, host time is device time)
    g.v_us[0] = us_since(t_begin);
    g.v_us[1] = g.v_us[2] = 0;
    if (!exchange && local_rc) return local_rc;
    const auto t_exchange = std::chrono::steady_clock::now();
    const unsigned long long* d_src = d_set;
    unsigned long long n_src = slots, n_keys = c[1];
    if (exchange) {
        // each rank saw only its tiles' hits: all-gather the key lists and de-duplicate (a key may be found twice)
        unsigned long long *d_list = nullptr, *d_all = nullptr, *d_set2 = nullptr;
        const int W = g.world;
        // (the two small exchange words were allocated with the communicator: nothing can fail between here and the collective)
        unsigned long long word[2] = {local_rc ? 0ull : n_keys, (unsigned long long)(unsigned)(local_rc ? 1 : 0)};
        std::vector<unsigned long long> words(2 * (size_t)W);
        auto agree = [&](const char* what, int own_rc) -> int {  // all-gather (count, status); a failure anywhere -> everyone leaves
            if (g.host_exchange) {  // group without RCCL: the words meet in host memory
                HxGuard hx;
                HX_BARRIER(W);      // (everybody is done with the previous round's slots)
                g_hx.words[(size_t)g.rank].assign(word, word + 2);
                HX_BARRIER(W);
                for (int r = 0; r < W; ++r) {
                    words[2 * (size_t)r] = g_hx.words[(size_t)r][0];
                    words[2 * (size_t)r + 1] = g_hx.words[(size_t)r][1];
                }
                hx.done = true;
            } else {
                if (!g.comm_ready) return fail(HVD_ERR_STATE, "no communicator on context %d (aborted after another rank's failure?)", g.id);
                HIP_TRY(hipMemcpyAsync(g.x_cnt_in, word, 16, hipMemcpyHostToDevice, g.stream));
                NCCL_TRY(ncclAllGather(g.x_cnt_in, g.x_cnt_all, 2, ncclUint64, g.comm, g.stream));
                HIP_TRY(hipMemcpyAsync(words.data(), g.x_cnt_all, 16 * (size_t)W, hipMemcpyDeviceToHost, g.stream));
                HIP_TRY(hipStreamSynchronize(g.stream));
            }
            for (int r = 0; r < W; ++r)
                if (words[2 * (size_t)r + 1]) {
                    t_agreed_exit = true;       // every rank reads the same words and leaves here, in lock-step
                    if (own_rc) return own_rc;  // our own failure: its message is already recorded
                    return fail(HVD_ERR_RCCL, "video search abandoned: rank %d failed %s", r, what);
                }
            return HVD_OK;
        };
        if (int rc = agree("before the key exchange", local_rc)) return rc;
        unsigned long long mx = 1, total = 0;
        for (int r = 0; r < W; ++r) {
            mx = std::max(mx, words[2 * (size_t)r]);
            total += words[2 * (size_t)r];
        }
        // the exchange buffers depend on the gathered counts: allocate, then agree once more before the big all-gather
        const int alloc_rc = [&]() -> int {
            SCR(S_LIST, 8 * mx, d_list);  // this rank's keys, padded with empty keys to the longest list
            SCR(S_LISTALL, 8 * mx * (size_t)W, d_all);
            return HVD_OK;
        }();
        word[0] = 0;
        word[1] = alloc_rc ? 1ull : 0ull;
        if (int rc = agree("while allocating the exchange buffers", alloc_rc)) return rc;
        HIP_TRY(hipMemsetAsync(d_list, 0xFF, 8 * mx, g.stream));
        HIP_TRY(hipMemsetAsync(d_counters + 2, 0, 8, g.stream));
        HIP_TRY(hvd::launch_set_to_list(d_set, slots, d_list, mx, d_counters + 2, g.stream));
        if (g.host_exchange) {  // every rank's list through host memory, the concatenation back to every device
            std::vector<unsigned long long>& mine = g_hx.words[(size_t)g.rank];
            HxGuard hx;  // (a failure between the barriers must not strand the peers: ADVICE r4)
            HX_BARRIER(W);
            mine.resize((size_t)mx);
            HIP_TRY(hipMemcpyAsync(mine.data(), d_list, 8 * (size_t)mx, hipMemcpyDeviceToHost, g.stream));
            HIP_TRY(hipStreamSynchronize(g.stream));
            HX_BARRIER(W);
            for (int r = 0; r < W; ++r)
                HIP_TRY(hipMemcpyAsync(d_all + (size_t)r * mx, g_hx.words[(size_t)r].data(), 8 * (size_t)mx, hipMemcpyHostToDevice, g.stream));
            HIP_TRY(hipStreamSynchronize(g.stream));
            HX_BARRIER(W);  // (the slots are free again only when everybody has copied them)
            hx.done = true;
        } else {
            if (!g.comm_ready) return fail(HVD_ERR_STATE, "no communicator on context %d (aborted after another rank's failure?)", g.id);
            NCCL_TRY(ncclAllGather(d_list, d_all, 8 * mx, ncclUint8, g.comm, g.stream));
        }
        unsigned long long slots2 = pow2_at_least(std::max<unsigned long long>(1ull << 16, 4ull * total));
        if (g.v_force_slots_log2) slots2 = 1ull << g.v_force_slots_log2;
        for (;;) {
            SCR(S_SET2, 8 * slots2, d_set2);
            HIP_TRY(hipMemsetAsync(d_set2, 0xFF, 8 * slots2, g.stream));
            HIP_TRY(hipMemsetAsync(d_counters, 0, 32, g.stream));
            HIP_TRY(hvd::launch_list_to_set(d_all, mx * (unsigned long long)W, d_set2, slots2 - 1, d_counters, g.stream));
            if (int rc = read_counters(d_counters, c)) return rc;
            if (c[0] == 0) break;
            slots2 *= 4;
        }
        d_src = d_set2;
        n_src = slots2;
        n_keys = c[1];
        g.v_us[1] = us_since(t_exchange);
    }
    const auto t_fold = std::chrono::steady_clock::now();
    unsigned long long pslots = pow2_at_least(std::max<unsigned long long>(1024, 4ull * n_keys));
    if (g.v_force_slots_log2) pslots = 1ull << g.v_force_slots_log2;
    for (;;) {
        unsigned long long* d_pkeys = nullptr;
        void* d_pcnt = nullptr;
        SCR(S_PKEYS, 8 * pslots, d_pkeys);
        SCR(S_PCNT, 8 * pslots, d_pcnt);
        HIP_TRY(hipMemsetAsync(d_pkeys, 0xFF, 8 * pslots, g.stream));
        HIP_TRY(hipMemsetAsync(d_pcnt, 0, 8 * pslots, g.stream));
        HIP_TRY(hipMemsetAsync(d_counters, 0, 32, g.stream));
        HIP_TRY(hvd::launch_keys_to_pairs(d_src, n_src, v.d_vid_q, v.d_vid_t, v.rect, d_pkeys, d_pcnt, pslots - 1, d_counters,
                                          g.stream));
        if (int rc = read_counters(d_counters, c)) return rc;
        if (c[0] == 0) break;
        pslots *= 4;
    }
    g.v_pslots = pslots;
    g.v_us[2] = us_since(t_fold);
    return HVD_OK;
}

// Pair map -> hvd_vmatch records (unordered) in d_out[cap]; *d_count (device uint64) = number of video pairs.
int vmatch_emit(hvd_vmatch* d_out, int64_t cap, unsigned long long* d_count) {
    HIP_TRY(hipMemsetAsync(d_count, 0, 8, g.stream));
    HIP_TRY(hvd::launch_pairs_emit((const unsigned long long*)g.scr[Ctx::S_PKEYS], g.scr[Ctx::S_PCNT], g.v_pslots, d_out,
                                   (unsigned long long)cap, d_count, g.stream));
    return HVD_OK;
}

bool vmatch_less(const hvd_vmatch& x, const hvd_vmatch& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; }

// build + emit into the pool's record buffer, grown until everything fits (only the emit is repeated)
int vmatch_to_host(const VmArgs& v, int64_t expect, std::vector<hvd_vmatch>& res) {
    if (int rc = vmatch_build(v)) return rc;
    unsigned long long* d_counters = nullptr;
    SCR(S_COUNTERS, 64, d_counters);
    int64_t dcap = std::max<int64_t>(1 << 12, expect);
    for (;;) {
        hvd_vmatch* d_out = nullptr;
        SCR(S_VOUT, sizeof(hvd_vmatch) * (size_t)dcap, d_out);
        if (int rc = vmatch_emit(d_out, dcap, d_counters + 3)) return rc;
        unsigned long long cnt = 0;
        HIP_TRY(hipMemcpyAsync(&cnt, d_counters + 3, 8, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        if ((int64_t)cnt > dcap) {
            dcap = (int64_t)cnt;
            continue;
        }
        res.resize((size_t)cnt);
        if (cnt) {
            HIP_TRY(hipMemcpyAsync(res.data(), d_out, sizeof(hvd_vmatch) * (size_t)cnt, hipMemcpyDeviceToHost, g.stream));
            HIP_TRY(hipStreamSynchronize(g.stream));
        }
        break;
    }
    std::sort(res.begin(), res.end(), vmatch_less);
    return HVD_OK;
}

int check_offsets(const int64_t* offsets, int64_t V, int64_t* nf) {
    if (V < 0 || !offsets) return fail(HVD_ERR_ARG, "bad offsets");
    if (offsets[0] != 0) return fail(HVD_ERR_ARG, "offsets[0] must be 0");
    for (int64_t v = 0; v < V; ++v)
        if (offsets[v + 1] < offsets[v]) return fail(HVD_ERR_ARG, "offsets must be non-decreasing");
    *nf = V > 0 ? offsets[V] : 0;
    if (*nf >= (1ll << 32) - 1 || V >= (1ll << 31)) return fail(HVD_ERR_ARG, "too many frames/videos");
    return HVD_OK;
}

// upload one side of a host library: frame hashes -> FP4 image, CSR offsets -> frame->video map
int upload_library(const uint8_t* frames, const int64_t* offsets, int64_t V, int64_t nf, Ctx::Scr s_db, Ctx::Scr s_img,
                   Ctx::Scr s_vid, void** d_img, int32_t** d_vid) {
    void* d_db = nullptr;
    long long* d_off = nullptr;
    if (int rc = scratch(s_db, 32 * (size_t)nf, &d_db)) return rc;
    size_t ib = 0;
    if (int rc = hvd_fp4_image_bytes(nf, &ib)) return rc;
    if (int rc = scratch(s_img, ib, d_img)) return rc;
    if (int rc = scratch(s_vid, 4 * (size_t)nf, (void**)d_vid)) return rc;
    SCR(S_OFF, 8 * (size_t)(V + 1), d_off);
    HIP_TRY(hipMemcpyAsync(d_db, frames, 32 * (size_t)nf, hipMemcpyHostToDevice, g.stream));
    HIP_TRY(hipMemcpyAsync(d_off, offsets, 8 * (size_t)(V + 1), hipMemcpyHostToDevice, g.stream));
    if (int rc = hvd_dev_expand_fp4(d_db, nf, *d_img)) return rc;
    HIP_TRY(hvd::launch_video_of_frames(d_off, (uint32_t)V, (unsigned long long)nf, *d_vid, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));  // S_OFF is reused by the other side
    return HVD_OK;
}

}  // namespace

extern "C" {

int hvd_vpdq_match_videos(const uint8_t* frames, const int64_t* offsets, int64_t V, int max_dist, hvd_vmatch* out,
                          int64_t cap, int64_t* out_count) {
    if (int rc = need_ready()) return rc;
    if (!out_count || cap < 0 || (cap > 0 && !out)) return fail(HVD_ERR_ARG, "bad arguments");
    if (max_dist < 0 || max_dist > 256) return fail(HVD_ERR_ARG, "max_dist=%d out of range [0,256]", max_dist);
    *out_count = 0;
    int64_t nf = 0;
    if (int rc = check_offsets(offsets, V, &nf)) return rc;
    if (nf < 2) return HVD_OK;
    if (!frames) return fail(HVD_ERR_ARG, "frames is NULL");
    std::vector<hvd_vmatch> res;
    if (g_nctx > 1 && max_dist < 128 && nf >= 4096) {
        // the group: library replicated on every device, tile (rb, cb) -> context (rb + cb) % W, key sets exchanged inside
        // vmatch_build (RCCL all-gather between the devices, host memory where the group has no RCCL); every rank ends up
        // with the whole result, rank 0's is returned
        const int W = g_nctx;
        int rc = run_on_group([&](int r) -> int {
            std::lock_guard<std::recursive_mutex> lk(g.h_mu);
            void* d_img = nullptr;
            int32_t* d_vid = nullptr;
            const int up = upload_library(frames, offsets, V, nf, Ctx::S_DB, Ctx::S_IMG, Ctx::S_VIDQ, &d_img, &d_vid);
            VmArgs v{d_img, (uint32_t)nf, d_img, (uint32_t)nf, false, d_vid, d_vid, nullptr, nullptr, max_dist, r, W};
            v.pre_rc = up;
            std::vector<hvd_vmatch> mine;
            if (int rc_ = vmatch_to_host(v, V, mine)) return rc_;
            if (r == 0) res.swap(mine);
            return HVD_OK;
        });
        if (rc) return rc;
        *out_count = (int64_t)res.size();
        if ((int64_t)res.size() > cap)
            return fail(HVD_ERR_OVERFLOW, "video match buffer too small: need %lld, cap %lld", (long long)res.size(),
                        (long long)cap);
        if (!res.empty()) memcpy(out, res.data(), sizeof(hvd_vmatch) * res.size());
        return HVD_OK;
    }
    std::lock_guard<std::recursive_mutex> lk(g.h_mu);
    if (max_dist >= 128) {
        // popcount route (a tolerance the reference never uses): frame-level hits reduced on the host
        std::vector<int32_t> vid((size_t)nf);
        for (int64_t v = 0; v < V; ++v)
            for (int64_t f = offsets[v]; f < offsets[v + 1]; ++f) vid[(size_t)f] = (int32_t)v;
        std::vector<hvd_pair> recs;
        int64_t fcap = std::max<int64_t>(1 << 16, nf), fcount = 0;
        for (;;) {
            if (int rc = allpairs_host_raw(frames, nf, vid.data(), max_dist, recs, fcap, &fcount)) return rc;
            if (fcount <= fcap) break;
            fcap = fcount;
        }
        aggregate_video_hits(recs, vid.data(), vid.data(), res);
    } else {
        void* d_img = nullptr;
        int32_t* d_vid = nullptr;
        if (int rc = upload_library(frames, offsets, V, nf, Ctx::S_DB, Ctx::S_IMG, Ctx::S_VIDQ, &d_img, &d_vid)) return rc;
        VmArgs v{d_img, (uint32_t)nf, d_img, (uint32_t)nf, false, d_vid, d_vid, nullptr, nullptr, max_dist, 0, 1};
        if (int rc = vmatch_to_host(v, V, res)) return rc;
    }
    *out_count = (int64_t)res.size();
    if ((int64_t)res.size() > cap)
        return fail(HVD_ERR_OVERFLOW, "video match buffer too small: need %lld, cap %lld", (long long)res.size(),
                    (long long)cap);
    if (!res.empty()) memcpy(out, res.data(), sizeof(hvd_vmatch) * res.size());
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
    std::vector<int32_t> gq, gt;
    if (ids_q) {  // frames of videos with equal ids are not compared (a query that is also in the target set)
        gq.resize((size_t)nq);
        gt.resize((size_t)nt);
        for (int64_t v = 0; v < VQ; ++v)
            for (int64_t f = offsets_q[v]; f < offsets_q[v + 1]; ++f) gq[(size_t)f] = ids_q[v];
        for (int64_t v = 0; v < VT; ++v)
            for (int64_t f = offsets_t[v]; f < offsets_t[v + 1]; ++f) gt[(size_t)f] = ids_t[v];
    }
    // one rank's share (rank r of W contexts; W = 1: the whole rectangle on the current context)
    auto one = [&](int r, int W, std::vector<hvd_vmatch>& res) -> int {
        std::lock_guard<std::recursive_mutex> lk(g.h_mu);
        void *d_iq = nullptr, *d_it = nullptr;
        int32_t *d_vq = nullptr, *d_vt = nullptr, *d_gq = nullptr, *d_gt = nullptr;
        auto upload = [&]() -> int {
            if (int rc = upload_library(frames_q, offsets_q, VQ, nq, Ctx::S_DB, Ctx::S_IMG, Ctx::S_VIDQ, &d_iq, &d_vq)) return rc;
            if (int rc = upload_library(frames_t, offsets_t, VT, nt, Ctx::S_DB2, Ctx::S_IMG2, Ctx::S_VIDT, &d_it, &d_vt)) return rc;
            if (ids_q) {
                SCR(S_GRP, 4 * (size_t)nq, d_gq);
                SCR(S_GRP2, 4 * (size_t)nt, d_gt);
                HIP_TRY(hipMemcpyAsync(d_gq, gq.data(), 4 * (size_t)nq, hipMemcpyHostToDevice, g.stream));
                HIP_TRY(hipMemcpyAsync(d_gt, gt.data(), 4 * (size_t)nt, hipMemcpyHostToDevice, g.stream));
                HIP_TRY(hipStreamSynchronize(g.stream));
            }
            return HVD_OK;
        };
        const int up = upload();
        if (W == 1 && up) return up;
        VmArgs v{d_iq, (uint32_t)nq, d_it, (uint32_t)nt, true, d_vq, d_vt, d_gq, d_gt, max_dist, r, W};
        v.pre_rc = up;
        return vmatch_to_host(v, VQ, res);
    };
    std::vector<hvd_vmatch> res;
    if (g_nctx > 1 && nq + nt >= 4096) {
        const int W = g_nctx;
        int rc = run_on_group([&](int r) -> int {
            std::vector<hvd_vmatch> mine;
            if (int rc_ = one(r, W, mine)) return rc_;
            if (r == 0) res.swap(mine);
            return HVD_OK;
        });
        if (rc) return rc;
    } else if (int rc = one(0, 1, res)) {
        return rc;
    }
    *out_count = (int64_t)res.size();
    if ((int64_t)res.size() > cap)
        return fail(HVD_ERR_OVERFLOW, "video match buffer too small: need %lld, cap %lld", (long long)res.size(),
                    (long long)cap);
    if (!res.empty()) memcpy(out, res.data(), sizeof(hvd_vmatch) * res.size());
    return HVD_OK;
}

/* ---- device-resident forms: hashes / images / maps already in HBM (BASELINE config 5) ---- */

int hvd_dev_video_of_frames(const void* d_offsets, int64_t V, int64_t n, void* d_out_video) {
    if (int rc = need_ready()) return rc;
    if (V < 0 || n < 0 || V >= (1ll << 31) || n >= (1ll << 32) - 1 || !d_offsets || (n > 0 && !d_out_video))
        return fail(HVD_ERR_ARG, "bad arguments");
    HIP_TRY(hvd::launch_video_of_frames((const long long*)d_offsets, (uint32_t)V, (unsigned long long)n, (int32_t*)d_out_video,
                                        g.stream));
    return HVD_OK;
}

int hvd_dev_compact_kept(const void* d_hashes, const void* d_quality, int64_t n, const void* d_offsets, int64_t V,
                         int min_quality, void* d_out_hashes, void* d_out_offsets, void* d_out_video, int64_t* out_kept) {
    if (int rc = need_ready()) return rc;
    if (n < 0 || V < 0 || n >= (1ll << 32) - 1 || V >= (1ll << 31) || !d_offsets || !d_out_offsets || !out_kept)
        return fail(HVD_ERR_ARG, "bad arguments");
    if (n > 0 && (!d_hashes || !d_quality || !d_out_hashes || !d_out_video)) return fail(HVD_ERR_ARG, "NULL device pointer");
    std::lock_guard<std::recursive_mutex> lk(g.h_mu);
    void* d_scr = nullptr;
    unsigned long long* d_counters = nullptr;
    SCR(S_COMPACT, hvd::compact_scratch_bytes((unsigned long long)n), d_scr);
    SCR(S_COUNTERS, 64, d_counters);
    HIP_TRY(hvd::launch_compact_kept(d_hashes, (const int32_t*)d_quality, (unsigned long long)n, (const long long*)d_offsets,
                                     (uint32_t)V, min_quality, d_out_hashes, (long long*)d_out_offsets, (int32_t*)d_out_video,
                                     d_scr, d_counters + 2, g.stream));
    unsigned long long kept = 0;
    HIP_TRY(hipMemcpyAsync(&kept, d_counters + 2, 8, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    *out_kept = (int64_t)kept;
    return HVD_OK;
}

int hvd_dev_vpdq_match_videos(const void* d_img, int64_t n, const void* d_video, int max_dist, int rank, int world,
                              void* d_out, int64_t cap, void* d_count) {
    if (int rc = need_ready()) return rc;
    if (n < 0 || n >= (1ll << 32) - 1) return fail(HVD_ERR_ARG, "n=%lld out of range", (long long)n);
    if (max_dist < 0 || max_dist >= 128) return fail(HVD_ERR_ARG, "max_dist=%d out of range [0,127]", max_dist);
    if (world < 1 || rank < 0 || rank >= world) return fail(HVD_ERR_ARG, "bad rank/world %d/%d", rank, world);
    if (cap < 0 || !d_count || (cap > 0 && !d_out)) return fail(HVD_ERR_ARG, "bad output buffer");
    std::lock_guard<std::recursive_mutex> lk(g.h_mu);
    if (n < 2) {
        HIP_TRY(hipMemsetAsync(d_count, 0, 8, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        return HVD_OK;
    }
    if (!d_img || !d_video) return fail(HVD_ERR_ARG, "d_img / d_video is NULL");
    VmArgs v{d_img, (uint32_t)n, d_img, (uint32_t)n, false, (const int32_t*)d_video, (const int32_t*)d_video, nullptr, nullptr,
             max_dist, rank, world};
    if (int rc = vmatch_build(v)) return rc;
    if (int rc = vmatch_emit((hvd_vmatch*)d_out, cap, (unsigned long long*)d_count)) return rc;
    HIP_TRY(hipStreamSynchronize(g.stream));
    return HVD_OK;
}

int hvd_dev_vpdq_emit_again(void* d_out, int64_t cap, void* d_count) {
    if (int rc = need_ready()) return rc;
    if (cap < 0 || !d_count || (cap > 0 && !d_out)) return fail(HVD_ERR_ARG, "bad output buffer");
    std::lock_guard<std::recursive_mutex> lk(g.h_mu);
    if (g.v_pslots == 0 || !g.scr[Ctx::S_PKEYS]) return fail(HVD_ERR_STATE, "no video search to emit from: call hvd_dev_vpdq_match_videos[_cross] first");
    if (int rc = vmatch_emit((hvd_vmatch*)d_out, cap, (unsigned long long*)d_count)) return rc;
    HIP_TRY(hipStreamSynchronize(g.stream));
    return HVD_OK;
}

int hvd_dev_vpdq_match_videos_cross(const void* d_img_q, int64_t nq, const void* d_video_q, const void* d_excl_q,
                                    const void* d_img_t, int64_t nt, const void* d_video_t, const void* d_excl_t,
                                    int max_dist, int rank, int world, void* d_out, int64_t cap, void* d_count) {
    if (int rc = need_ready()) return rc;
    if (nq < 0 || nt < 0 || nq >= (1ll << 32) - 1 || nt >= (1ll << 32) - 1) return fail(HVD_ERR_ARG, "set size out of range");
    if (max_dist < 0 || max_dist >= 128) return fail(HVD_ERR_ARG, "max_dist=%d out of range [0,127]", max_dist);
    if (world < 1 || rank < 0 || rank >= world) return fail(HVD_ERR_ARG, "bad rank/world %d/%d", rank, world);
    if (cap < 0 || !d_count || (cap > 0 && !d_out)) return fail(HVD_ERR_ARG, "bad output buffer");
    if ((d_excl_q == nullptr) != (d_excl_t == nullptr)) return fail(HVD_ERR_ARG, "pass both exclusion maps or neither");
    std::lock_guard<std::recursive_mutex> lk(g.h_mu);
    if (nq == 0 || nt == 0) {
        HIP_TRY(hipMemsetAsync(d_count, 0, 8, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        return HVD_OK;
    }
    if (!d_img_q || !d_img_t || !d_video_q || !d_video_t) return fail(HVD_ERR_ARG, "NULL image / video map");
    VmArgs v{d_img_q, (uint32_t)nq, d_img_t, (uint32_t)nt, true, (const int32_t*)d_video_q, (const int32_t*)d_video_t,
             (const int32_t*)d_excl_q, (const int32_t*)d_excl_t, max_dist, rank, world};
    if (int rc = vmatch_build(v)) return rc;
    if (int rc = vmatch_emit((hvd_vmatch*)d_out, cap, (unsigned long long*)d_count)) return rc;
    HIP_TRY(hipStreamSynchronize(g.stream));
    return HVD_OK;
}

#ifndef HVD_NO_BENCH_SYMBOLS
int hvd_dev_synth_video_frames(void* d_frames, int64_t v0, int64_t n_videos, int frames_per_video, uint64_t seed,
                               const void* d_copy_of) {
    if (int rc = need_ready()) return rc;
    if (v0 < 0 || n_videos < 0 || frames_per_video < 1 || n_videos * (int64_t)frames_per_video >= (1ll << 31))
        return fail(HVD_ERR_ARG, "bad synthetic library shape");
    if (n_videos == 0) return HVD_OK;
    if (!d_frames) return fail(HVD_ERR_ARG, "d_frames is NULL");
    HIP_TRY(hvd::launch_synth_frames64((uint8_t*)d_frames, v0, (uint32_t)frames_per_video,
                                       (unsigned long long)n_videos * (unsigned long long)frames_per_video, seed,
                                       (const int32_t*)d_copy_of, g.stream));
    return HVD_OK;
}

#endif  // HVD_NO_BENCH_SYMBOLS

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

static void free_exchange_buffers() {
    void** ps[] = {&g.x_cnt_in, &g.x_cnt_all, &g.x_send, &g.x_recv};
    for (void** p : ps) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    g.x_send_cap = g.x_recv_cap = 0;
}

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
