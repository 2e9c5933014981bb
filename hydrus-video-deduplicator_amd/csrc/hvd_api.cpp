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

#include "hvd_internal.h"
#include "../../include/hvd_mi355x_bench.h"

namespace hvdi {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

Ctx g_ctx[kMaxCtx];
int g_nctx = 0;
bool g_group_rccl = false;
bool g_group_was_rccl = false;
thread_local bool t_agreed_exit = false;
int g_match_server = 1;
thread_local int t_ctx = 0;
std::mutex g_mu;
HostExchange g_hx;

int need_ready() {
    if (t_ctx >= g_nctx) t_ctx = 0;  // (a thread that selected a context of an earlier, larger group)
    if (!g.ready) return fail(HVD_ERR_STATE, "hvd_init() has not been called (no CPU fallback exists)");
    // HIP's current device is per host thread; the reference calls this path from the main thread or
    // from a QThread worker (gui/gui.py:195-237), so every entry re-asserts the bound device.
    hipError_t e = hipSetDevice(g.device);
    if (e != hipSuccess) return fail(HVD_ERR_HIP, "hipSetDevice(%d): %s", g.device, hipGetErrorString(e));
    return HVD_OK;
}

}  // namespace hvdi

using namespace hvdi;

namespace {

// Multi-process GPU work on hosts whose driver only supports dmabuf IPC needs HSA_ENABLE_IPC_MODE_LEGACY=0 (without it RCCL's
// hipIpcGetMemHandle fails with "invalid argument"). The variable is read when the HSA runtime starts, i.e. at the first HIP
// call: set it when the library is loaded, never over a value the user chose.
__attribute__((constructor)) void hvd_default_ipc_mode() { setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", /*overwrite=*/0); }

// pdqhashing.cpp fill_dct_matrix_64_cached: float scale * double cos, rounded once.
void fill_dct(float* out) {
    const float scale = (float)std::sqrt(2.0 / 64.0);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 64; ++j)
            out[i * 64 + j] = (float)((double)scale * std::cos((M_PI / 2 / 64.0) * (i + 1) * (2 * j + 1)));
}

}  // namespace


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

namespace hvdi {
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

}  // extern "C"

namespace hvdi {
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
