// hvd_search.cpp -- the host-buffer entry points of the C-ABI (what the vpdq-shaped shim binds: hash frames, one pair, all pairs,
// video-level search; each fans out over the device group by itself) and the video-level search on the device (K3: key set,
// agreement step, key exchange, fold, emit). Split out of hvd_api.cpp in round 6; shared state: hvd_internal.h.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "hvd_internal.h"
#include "../../include/hvd_mi355x_bench.h"

using namespace hvdi;

namespace {
constexpr unsigned long long kMatchServerIdleUs = 300;   // the server leaves after this long without a call ...
constexpr unsigned long long kMatchServerLifeUs = 2000;  // ... and after this long in any case (another thread's hipFree / device-wide wait gets its turn)
}  // namespace

extern "C" {


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

}  // extern "C"
