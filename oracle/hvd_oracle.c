/*
 * hvd_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT).
 *
 * Plain-C restatement of the arithmetic that hydrus-video-deduplicator delegates to
 * the un-vendored third-party wheel `hvdaccelerators==0.4.0` (reference pyproject.toml:36,
 * uv.lock:186-189), which wraps Meta ThreatExchange `pdq` + `vpdq` (docs/credits.md:7-9).
 *
 * PARITY UNPINNED: neither the wheel nor the reference's golden vectors (the
 * tests/testdb git submodule, .gitmodules:1-3) exist in the build container, so
 * this file restates the *published* PDQ / vPDQ algorithm (ThreatExchange
 * pdq/cpp/hashing/pdqhashing.cpp, pdq/cpp/downscaling/downscaling.cpp,
 * pdq/cpp/hashing/torben.cpp, vpdq/cpp/vpdq/cpp/matchTwoHash) and anchors on the
 * reference's own call sites:
 *   - VideoHasher.hash_frame(bytes(rgb24 plane))       vpdqpy/vpdqpy.py:113-119
 *   - vpdq.matchHash(q, t, 31) / matchHashBytes(a,b,31) vpdqpy/vpdqpy.py:56, db/vptree.py:31
 *   - 32 bytes per frame hash, little-endian w[0..15]   dedup.py:83, db/DedupeDB.py:535-559
 *   - frames with quality < 31 are dropped               db/DedupeDB.py:550-553
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library. The product (libhvd_mi355x.so) never links or calls it.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math; no -march=native
 * so the prebuilt .so also runs on the GPU box's host CPU).
 *
 * Floating-point contract (what "bit-exact" means for the GPU path):
 *   every float op below is a separately rounded IEEE-754 binary32 op in the
 *   written order (x86-64 baseline wheels have no FMA contraction); the DCT
 *   matrix is computed in double and rounded once to float.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HVD_OK 0
#define HVD_ERR_ARG (-1)
#define HVD_ERR_OVERFLOW (-3)

typedef struct {
    uint32_t i, j, dist, pad;
} hvd_pair;

typedef struct {
    uint32_t a, b, q_hits, t_hits;
} hvd_vmatch;

/* ---------------------------------------------------------------- luma -- */

/* pdqhashing.cpp fillFloatLumaFromRGB: float coefficients, left-to-right sum. */
static const float kLumaR = 0.299f, kLumaG = 0.587f, kLumaB = 0.114f;

void hvd_cpu_luma_rgb24(const uint8_t* rgb, int64_t npix, float* luma) {
    for (int64_t p = 0; p < npix; ++p) {
        float r = (float)rgb[3 * p + 0], g = (float)rgb[3 * p + 1], b = (float)rgb[3 * p + 2];
        float y = kLumaR * r;
        y = y + kLumaG * g;
        y = y + kLumaB * b;
        luma[p] = y;
    }
}

/* Gray entry is *defined* as the RGB entry with R=G=B (SURVEY 7 "hard parts"),
 * so VideoHasher.hash_frame(rgb with r=g=b) and the gray batch entry agree. */
void hvd_cpu_luma_gray(const uint8_t* g, int64_t npix, float* luma) {
    for (int64_t p = 0; p < npix; ++p) {
        float v = (float)g[p];
        float y = kLumaR * v;
        y = y + kLumaG * v;
        y = y + kLumaB * v;
        luma[p] = y;
    }
}

/* ------------------------------------------------------- Jarosz filter -- */

/* downscaling.cpp computeJaroszFilterWindowSize(old, new=64). */
static int jarosz_window(int old_dim) { return (old_dim + 2 * 64 - 1) / (2 * 64); }

/* downscaling.cpp box1DFloat: sequential running sum, 4-phase edge schedule. */
static void box1d(const float* in, float* out, int n, int stride, int w) {
    int half = (w + 2) / 2;
    int p1 = half - 1, p2 = w - half + 1, p3 = n - w, p4 = half - 1;
    int li = 0, ri = 0, oi = 0;
    float sum = 0.0f;
    int cur = 0;
    for (int i = 0; i < p1; ++i) {
        sum += in[ri];
        cur++;
        ri += stride;
    }
    for (int i = 0; i < p2; ++i) {
        sum += in[ri];
        cur++;
        out[oi] = sum / (float)cur;
        ri += stride;
        oi += stride;
    }
    for (int i = 0; i < p3; ++i) {
        sum += in[ri];
        sum -= in[li];
        out[oi] = sum / (float)cur;
        li += stride;
        ri += stride;
        oi += stride;
    }
    for (int i = 0; i < p4; ++i) {
        sum -= in[li];
        cur--;
        out[oi] = sum / (float)cur;
        li += stride;
        oi += stride;
    }
}

/* downscaling.cpp jaroszFilterFloat: nreps x (rows then cols); result ends in buf1. */
static void jarosz(float* buf1, float* buf2, int h, int w, int win_rows, int win_cols, int nreps) {
    for (int r = 0; r < nreps; ++r) {
        for (int i = 0; i < h; ++i) box1d(buf1 + (size_t)i * w, buf2 + (size_t)i * w, w, 1, win_rows);
        for (int j = 0; j < w; ++j) box1d(buf2 + j, buf1 + j, h, w, win_cols);
    }
}

/* downscaling.cpp decimateFloat: sample pixel centres, double arithmetic. */
static void decimate64(const float* in, int h, int w, float* out /*64*64*/) {
    for (int i = 0; i < 64; ++i) {
        int ini = (int)(((i + 0.5) * h) / 64);
        for (int j = 0; j < 64; ++j) {
            int inj = (int)(((j + 0.5) * w) / 64);
            out[i * 64 + j] = in[(size_t)ini * w + inj];
        }
    }
}

/* --------------------------------------------------------- quality ------ */

/* pdqhashing.cpp computePDQImageDomainQualityMetric. */
static int quality64(const float* a /*64*64*/) {
    int gsum = 0;
    for (int i = 0; i < 63; ++i)
        for (int j = 0; j < 64; ++j) {
            float u = a[i * 64 + j], v = a[(i + 1) * 64 + j];
            int d = (int)(((u - v) * 100.0f) / 255.0f);
            gsum += abs(d);
        }
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 63; ++j) {
            float u = a[i * 64 + j], v = a[i * 64 + j + 1];
            int d = (int)(((u - v) * 100.0f) / 255.0f);
            gsum += abs(d);
        }
    int q = gsum / 90;
    return q > 100 ? 100 : q;
}

/* ------------------------------------------------------------- DCT ------ */

static float g_dct[16 * 64];
static pthread_once_t g_dct_once = PTHREAD_ONCE_INIT;

/* pdqhashing.cpp fill_dct_matrix_64_cached: float scale * double cos, stored as float. */
static void dct_init(void) {
    const float scale = (float)sqrt(2.0 / 64.0);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 64; ++j)
            g_dct[i * 64 + j] = (float)((double)scale * cos((M_PI / 2 / 64.0) * (i + 1) * (2 * j + 1)));
}

const float* hvd_cpu_dct_matrix(void) {
    pthread_once(&g_dct_once, dct_init);
    return g_dct;
}

/* DCT accumulation mode. 0 ("strict", default): `sumk += D*A` as a separately rounded multiply and
 * add -- what upstream's x86-64 wheels compute (baseline x86-64 has no FMA to contract into).
 * 1 ("fma"): the same statement contracted into one fused multiply-add per step, which is what
 * clang emits for it on arm64 (-ffp-contract=on is its default) -- upstream's macOS arm64 wheels
 * (uv.lock:186-206 lists them). The GPU library offers the same two modes. */
static int g_dct_fma = 0;
void hvd_cpu_set_dct_mode(int fma) { g_dct_fma = fma != 0; }
int hvd_cpu_get_dct_mode(void) { return g_dct_fma; }

/* pdqhashing.cpp dct64To16: T = D*A (16x64), B = T*D^T (16x16); k-sequential. */
static void dct64to16(const float* A, float* T, float* B) {
    const float* D = hvd_cpu_dct_matrix();
    const int fma = g_dct_fma;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 64; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 64; ++k) {
                if (fma) {
                    s = fmaf(D[i * 64 + k], A[k * 64 + j], s);
                } else {
                    float p = D[i * 64 + k] * A[k * 64 + j];
                    s = s + p;
                }
            }
            T[i * 64 + j] = s;
        }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float s = 0.0f;
            for (int k = 0; k < 64; ++k) {
                if (fma) {
                    s = fmaf(T[i * 64 + k], D[j * 64 + k], s);
                } else {
                    float p = T[i * 64 + k] * D[j * 64 + k];
                    s = s + p;
                }
            }
            B[i * 16 + j] = s;
        }
}

/* ---------------------------------------------------------- median ------ */

/* torben.cpp (N. Devillard's public-domain Torben median), n = 256. */
static float torben(const float* m, int n) {
    int less, greater, equal;
    float mn, mx, guess, maxlt, mingt;
    mn = mx = m[0];
    for (int i = 1; i < n; ++i) {
        if (m[i] < mn) mn = m[i];
        if (m[i] > mx) mx = m[i];
    }
    for (;;) {
        guess = (mn + mx) / 2;
        less = greater = equal = 0;
        maxlt = mn;
        mingt = mx;
        for (int i = 0; i < n; ++i) {
            if (m[i] < guess) {
                less++;
                if (m[i] > maxlt) maxlt = m[i];
            } else if (m[i] > guess) {
                greater++;
                if (m[i] < mingt) mingt = m[i];
            } else
                equal++;
        }
        if (less <= (n + 1) / 2 && greater <= (n + 1) / 2) break;
        if (less > greater)
            mx = maxlt;
        else
            mn = mingt;
    }
    if (less >= (n + 1) / 2) return maxlt;
    if (less + equal >= (n + 1) / 2) return guess;
    return mingt;
}

/* pdqhashing.cpp pdqBuffer16x16ToBits + Hash256::setBit: bit k=i*16+j lives in
 * uint16 word k>>4, bit k&15; the reference's BLOB is the little-endian image of
 * w[0..15] (db/DedupeDB.py:538-544,553) => byte k>>3, bit k&7. */
static void bits_from_dct(const float* B, uint8_t hash[32]) {
    float med = torben(B, 256);
    memset(hash, 0, 32);
    for (int k = 0; k < 256; ++k)
        if (B[k] > med) hash[k >> 3] |= (uint8_t)(1u << (k & 7));
}

/* ------------------------------------------------- one frame, from luma -- */

/* pdqhashing.cpp pdqHash256FromFloatLuma. luma is clobbered; scratch has h*w floats.
 * coeffs (nullable) receives the 16x16 DCT output for cross-implementation checks. */
int hvd_cpu_pdq_from_luma(float* luma, float* scratch, int h, int w, uint8_t hash[32], int32_t* quality,
                          float* coeffs) {
    float a64[64 * 64], T[16 * 64], B[16 * 16];
    if (h < 64 || w < 64) return HVD_ERR_ARG;
    if (h == 64 && w == 64) {
        memcpy(a64, luma, sizeof a64); /* upstream: already-downsampled video frames skip the blur */
    } else {
        jarosz(luma, scratch, h, w, jarosz_window(w), jarosz_window(h), 2);
        decimate64(luma, h, w, a64);
    }
    *quality = quality64(a64);
    dct64to16(a64, T, B);
    bits_from_dct(B, hash);
    if (coeffs) memcpy(coeffs, B, sizeof B);
    return HVD_OK;
}

/* ------------------------------------------------------- frame batches -- */

typedef struct {
    const uint8_t* frames;
    int64_t begin, end;
    int h, w, channels;
    uint8_t* hashes;
    int32_t* quality;
    float* coeffs;
    int rc;
} frame_job;

static void* frame_worker(void* arg) {
    frame_job* jb = (frame_job*)arg;
    size_t npix = (size_t)jb->h * jb->w;
    float* luma = (float*)malloc(npix * sizeof(float));
    float* scratch = (float*)malloc(npix * sizeof(float));
    jb->rc = HVD_OK;
    for (int64_t f = jb->begin; f < jb->end; ++f) {
        const uint8_t* src = jb->frames + (size_t)f * npix * jb->channels;
        if (jb->channels == 3)
            hvd_cpu_luma_rgb24(src, (int64_t)npix, luma);
        else
            hvd_cpu_luma_gray(src, (int64_t)npix, luma);
        int rc = hvd_cpu_pdq_from_luma(luma, scratch, jb->h, jb->w, jb->hashes + 32 * f, jb->quality + f,
                                       jb->coeffs ? jb->coeffs + 256 * f : NULL);
        if (rc != HVD_OK) jb->rc = rc;
    }
    free(luma);
    free(scratch);
    return NULL;
}

static int hash_frames(const uint8_t* frames, int64_t n, int h, int w, int channels, uint8_t* hashes,
                       int32_t* quality, float* coeffs, int num_threads) {
    if (n < 0 || h < 64 || w < 64 || (n > 0 && (!frames || !hashes || !quality))) return HVD_ERR_ARG;
    if (num_threads < 1) num_threads = 1;
    if (num_threads > 256) num_threads = 256;
    if ((int64_t)num_threads > n) num_threads = n > 0 ? (int)n : 1;
    hvd_cpu_dct_matrix();
    frame_job jobs[256];
    pthread_t th[256];
    int rc = HVD_OK;
    for (int t = 0; t < num_threads; ++t) {
        jobs[t] = (frame_job){frames, n * t / num_threads, n * (t + 1) / num_threads, h, w, channels,
                              hashes, quality, coeffs, HVD_OK};
        if (num_threads == 1)
            frame_worker(&jobs[t]);
        else
            pthread_create(&th[t], NULL, frame_worker, &jobs[t]);
    }
    for (int t = 0; t < num_threads; ++t) {
        if (num_threads > 1) pthread_join(th[t], NULL);
        if (jobs[t].rc != HVD_OK) rc = jobs[t].rc;
    }
    return rc;
}

/* Counterpart of VideoHasher.hash_frame for a batch (vpdqpy/vpdqpy.py:118). */
int hvd_cpu_pdq_hash_frames_gray_u8(const uint8_t* frames, int64_t n, int h, int w, uint8_t* out_hashes,
                                    int32_t* out_quality, float* out_coeffs, int num_threads) {
    return hash_frames(frames, n, h, w, 1, out_hashes, out_quality, out_coeffs, num_threads);
}

int hvd_cpu_pdq_hash_frames_rgb24_u8(const uint8_t* frames, int64_t n, int h, int w, uint8_t* out_hashes,
                                     int32_t* out_quality, float* out_coeffs, int num_threads) {
    return hash_frames(frames, n, h, w, 3, out_hashes, out_quality, out_coeffs, num_threads);
}

/* ------------------------------------------------------------- Hamming -- */

/* pdq Hash256::hammingDistance: popcount of the XOR over 256 bits. */
int hvd_cpu_hamming256(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

typedef struct {
    const uint8_t* db;
    const int32_t* group;
    int64_t n, row_begin, row_end;
    int max_dist;
    hvd_pair* out; /* private buffer */
    int64_t cap, count;
    const uint64_t* tdb; /* AVX-512 path: the DB transposed in blocks of 8 (NULL: scalar loop) */
} pair_job;

/* The same scan with AVX-512 VPOPCNTDQ where the host has it (runtime CPUID check; HVD_ORACLE_NO_AVX512=1 forces the
 * scalar loop). The DB is transposed once per call into blocks of 8 hashes (qword k of the 8 hashes = one 512-bit
 * vector), so a block costs 4 loads, 4 xors with the broadcast query words, 4 vpopcntq, 3 adds and one compare for 8
 * comparisons, with no horizontal reduction. It only FINDS the rare candidates faster: a flagged candidate is then
 * handled by the very scalar statement the plain loop uses, in the same order, so counts and records are identical.
 * This is the CPU baseline a current x86 host deserves (-march=native, VPOPCNTDQ), not a different
 * algorithm. */
#if defined(__x86_64__)
#include <immintrin.h>
static int g_avx512 = -1;
static int have_avx512_vpopcnt(void) {
    if (g_avx512 < 0) {
        const char* off = getenv("HVD_ORACLE_NO_AVX512");
        __builtin_cpu_init();
        g_avx512 = (!(off && *off && *off != '0') && __builtin_cpu_supports("avx512f") &&
                    __builtin_cpu_supports("avx512vpopcntdq")) ? 1 : 0;
    }
    return g_avx512;
}
/* 8-bit mask of the hashes of block blk (8 hashes, transposed at t) within max_dist of a */
__attribute__((target("avx512f,avx512vpopcntdq"), always_inline)) static inline unsigned block_mask_avx512(
    const uint64_t* t, __m512i a0, __m512i a1, __m512i a2, __m512i a3, __m512i lim) {
    __m512i c0 = _mm512_popcnt_epi64(_mm512_xor_si512(a0, _mm512_load_si512((const void*)(t))));
    __m512i c1 = _mm512_popcnt_epi64(_mm512_xor_si512(a1, _mm512_load_si512((const void*)(t + 8))));
    __m512i c2 = _mm512_popcnt_epi64(_mm512_xor_si512(a2, _mm512_load_si512((const void*)(t + 16))));
    __m512i c3 = _mm512_popcnt_epi64(_mm512_xor_si512(a3, _mm512_load_si512((const void*)(t + 24))));
    return _mm512_cmple_epi64_mask(_mm512_add_epi64(_mm512_add_epi64(c0, c1), _mm512_add_epi64(c2, c3)), lim);
}
/* first block in [blk, nblk) holding a candidate for query a (its mask in *mask), or nblk */
__attribute__((target("avx512f,avx512vpopcntdq"))) static int64_t scan_blocks_avx512(const uint64_t* tdb, int64_t blk,
                                                                                     int64_t nblk, const uint64_t* a,
                                                                                     int max_dist, unsigned* mask) {
    const __m512i a0 = _mm512_set1_epi64((long long)a[0]), a1 = _mm512_set1_epi64((long long)a[1]);
    const __m512i a2 = _mm512_set1_epi64((long long)a[2]), a3 = _mm512_set1_epi64((long long)a[3]);
    const __m512i lim = _mm512_set1_epi64(max_dist);
    for (; blk + 2 <= nblk; blk += 2) {
        unsigned m0 = block_mask_avx512(tdb + 32 * blk, a0, a1, a2, a3, lim);
        unsigned m1 = block_mask_avx512(tdb + 32 * blk + 32, a0, a1, a2, a3, lim);
        if (m0 | m1) {
            *mask = m0 ? m0 : m1;
            return m0 ? blk : blk + 1;
        }
    }
    for (; blk < nblk; ++blk) {
        unsigned m0 = block_mask_avx512(tdb + 32 * blk, a0, a1, a2, a3, lim);
        if (m0) {
            *mask = m0;
            return blk;
        }
    }
    return nblk;
}
/* tdb[(blk*4 + k)*8 + lane] = qword k of hash blk*8 + lane, full blocks only; 64-byte aligned */
static uint64_t* transpose_blocks(const uint64_t* d, int64_t nblk) {
    uint64_t* t = NULL;
    if (nblk <= 0 || posix_memalign((void**)&t, 64, (size_t)nblk * 32 * sizeof(uint64_t)) != 0) return NULL;
    for (int64_t b = 0; b < nblk; ++b)
        for (int k = 0; k < 4; ++k)
            for (int l = 0; l < 8; ++l) t[(b * 4 + k) * 8 + l] = d[4 * (b * 8 + l) + k];
    return t;
}
#else
static int have_avx512_vpopcnt(void) { return 0; }
#endif

int hvd_cpu_allpairs_uses_avx512(void) { return have_avx512_vpopcnt(); }

/* THE comparison (pdq Hash256::hammingDistance + the tolerance test + the group filter): every path ends here. */
static inline void compare_and_emit(pair_job* jb, int64_t i, int64_t j, uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3) {
    const uint64_t* d = (const uint64_t*)jb->db;
    int dist = __builtin_popcountll(a0 ^ d[4 * j]) + __builtin_popcountll(a1 ^ d[4 * j + 1]) +
               __builtin_popcountll(a2 ^ d[4 * j + 2]) + __builtin_popcountll(a3 ^ d[4 * j + 3]);
    if (dist <= jb->max_dist) {
        if (jb->group && jb->group[i] == jb->group[j]) return;
        if (jb->count < jb->cap) jb->out[jb->count] = (hvd_pair){(uint32_t)i, (uint32_t)j, (uint32_t)dist, 0};
        jb->count++;
    }
}

static int pair_cmp(const void* x, const void* y) {
    const hvd_pair* a = (const hvd_pair*)x;
    const hvd_pair* b = (const hvd_pair*)y;
    if (a->i != b->i) return a->i < b->i ? -1 : 1;
    return a->j < b->j ? -1 : (a->j > b->j ? 1 : 0);
}

static void* pair_worker(void* arg) {
    pair_job* jb = (pair_job*)arg;
    const uint64_t* d = (const uint64_t*)jb->db; /* db is 8-byte aligned by contract of the callers */
    jb->count = 0;
#if defined(__x86_64__)
    if (jb->tdb) {
        /* Cache-blocked: the candidates are walked in chunks of kChunk hashes (256 KiB of the transposed copy: stays
         * in L2) and every query row of this thread meets a chunk before the next chunk is touched -- one pass over
         * the DB per THREAD instead of one per query row (at 1 M hashes the row-at-a-time scan streamed 32 MB per row
         * and ran at memory speed). Records come out chunk by chunk and are sorted by (i, j) afterwards. */
        enum { kChunk = 8192 };
        const int64_t nblk = jb->n / 8;
        for (int64_t c0 = 0; c0 < jb->n; c0 += kChunk) {
            const int64_t c1 = c0 + kChunk < jb->n ? c0 + kChunk : jb->n;
            for (int64_t i = jb->row_begin; i < jb->row_end && i + 1 < c1; ++i) {
                uint64_t a0 = d[4 * i], a1 = d[4 * i + 1], a2 = d[4 * i + 2], a3 = d[4 * i + 3];
                int64_t j = i + 1 > c0 ? i + 1 : c0;
                for (; j < c1 && (j & 7); ++j) compare_and_emit(jb, i, j, a0, a1, a2, a3); /* up to a block boundary */
                int64_t blk = j / 8;
                const int64_t blk_end = c1 / 8 < nblk ? c1 / 8 : nblk; /* full blocks inside this chunk */
                while (blk < blk_end) {
                    unsigned mask = 0;
                    blk = scan_blocks_avx512(jb->tdb, blk, blk_end, &d[4 * i], jb->max_dist, &mask);
                    if (blk >= blk_end) break;
                    for (; mask; mask &= mask - 1) compare_and_emit(jb, i, blk * 8 + __builtin_ctz(mask), a0, a1, a2, a3);
                    ++blk;
                }
                if (j < blk_end * 8) j = blk_end * 8;
                for (; j < c1; ++j) compare_and_emit(jb, i, j, a0, a1, a2, a3); /* the chunk's tail below a full block */
            }
        }
        if (jb->count > 1) qsort(jb->out, (size_t)(jb->count < jb->cap ? jb->count : jb->cap), sizeof(hvd_pair), pair_cmp);
        return NULL;
    }
#endif
    for (int64_t i = jb->row_begin; i < jb->row_end; ++i) {
        uint64_t a0 = d[4 * i], a1 = d[4 * i + 1], a2 = d[4 * i + 2], a3 = d[4 * i + 3];
        for (int64_t j = i + 1; j < jb->n; ++j) compare_and_emit(jb, i, j, a0, a1, a2, a3);
    }
    return NULL;
}

/* Brute-force ground truth of the pair predicate (finding 5):
 * all i<j with hamming(db[i], db[j]) <= max_dist (and group[i] != group[j] when
 * group is given). Output is sorted by (i, j). Rows [row_begin,row_end) only. */
static int allpairs_rows_impl(const uint8_t* db, int64_t n, const int32_t* group, int64_t row_begin, int64_t row_end,
                              int max_dist, hvd_pair* out, int64_t cap, int64_t* out_count, int num_threads,
                              uint64_t* tdb_shared) {
    if (n < 0 || !out_count || (n > 0 && !db) || row_begin < 0 || row_end > n || cap < 0) return HVD_ERR_ARG;
    if (((uintptr_t)db & 7) != 0) return HVD_ERR_ARG;
    if (num_threads < 1) num_threads = 1;
    if (num_threads > 256) num_threads = 256;
    int64_t rows = row_end > row_begin ? row_end - row_begin : 0;
    if (rows == 0) {
        *out_count = 0;
        return HVD_OK;
    }
    if (num_threads > rows) num_threads = (int)rows;
    pair_job jobs[256];
    pthread_t th[256];
    uint64_t* tdb = tdb_shared;
#if defined(__x86_64__)
    if (!tdb && have_avx512_vpopcnt() && n >= 64) tdb = transpose_blocks((const uint64_t*)db, n / 8);
#endif
    /* Split rows so that each thread gets ~equal triangle area. */
    double total = 0;
    for (int64_t i = row_begin; i < row_end; ++i) total += (double)(n - 1 - i);
    int64_t r = row_begin;
    double acc = 0;
    for (int t = 0; t < num_threads; ++t) {
        int64_t b = r;
        double want = total * (t + 1) / num_threads;
        while (r < row_end && (acc < want || t == num_threads - 1)) {
            acc += (double)(n - 1 - r);
            r++;
        }
        jobs[t] = (pair_job){db, group, n, b, r, max_dist, NULL, cap, 0, tdb};
        jobs[t].out = (hvd_pair*)malloc((size_t)(cap > 0 ? cap : 1) * sizeof(hvd_pair));
        if (num_threads == 1)
            pair_worker(&jobs[t]);
        else
            pthread_create(&th[t], NULL, pair_worker, &jobs[t]);
    }
    int64_t count = 0;
    for (int t = 0; t < num_threads; ++t) {
        if (num_threads > 1) pthread_join(th[t], NULL);
        for (int64_t k = 0; k < jobs[t].count && k < jobs[t].cap; ++k) {
            if (count + k < cap) out[count + k] = jobs[t].out[k];
        }
        count += jobs[t].count;
        free(jobs[t].out);
    }
    if (!tdb_shared) free(tdb);
    *out_count = count;
    return count > cap ? HVD_ERR_OVERFLOW : HVD_OK;
}

int hvd_cpu_allpairs_hamming256_rows(const uint8_t* db, int64_t n, const int32_t* group, int64_t row_begin,
                                     int64_t row_end, int max_dist, hvd_pair* out, int64_t cap,
                                     int64_t* out_count, int num_threads) {
    return allpairs_rows_impl(db, n, group, row_begin, row_end, max_dist, out, cap, out_count, num_threads, NULL);
}

/* Several row bands of the same brute force in one call (the full-size parity checks sample bands of a DB too large to
 * scan whole): bands[2k], bands[2k+1] = [row_begin, row_end) of band k, ascending and disjoint, so that the output --
 * band after band, each sorted by (i, j) -- is sorted as a whole. The block-transposed copy of the DB is built once. */
int hvd_cpu_allpairs_hamming256_bands(const uint8_t* db, int64_t n, const int32_t* group, const int64_t* bands,
                                      int64_t n_bands, int max_dist, hvd_pair* out, int64_t cap, int64_t* out_count,
                                      int num_threads) {
    if (n < 0 || !out_count || (n > 0 && !db) || n_bands < 0 || (n_bands > 0 && !bands) || cap < 0) return HVD_ERR_ARG;
    if (((uintptr_t)db & 7) != 0) return HVD_ERR_ARG;
    for (int64_t k = 0; k < n_bands; ++k)
        if (bands[2 * k] < (k ? bands[2 * k - 1] : 0) || bands[2 * k + 1] < bands[2 * k] || bands[2 * k + 1] > n) return HVD_ERR_ARG;
    uint64_t* tdb = NULL;
#if defined(__x86_64__)
    if (have_avx512_vpopcnt() && n >= 64) tdb = transpose_blocks((const uint64_t*)db, n / 8);
#endif
    int64_t count = 0;
    for (int64_t k = 0; k < n_bands; ++k) {
        int64_t c = 0;
        const int64_t room = cap > count ? cap - count : 0;
        int rc = allpairs_rows_impl(db, n, group, bands[2 * k], bands[2 * k + 1], max_dist, out + (room ? count : 0), room, &c,
                                    num_threads, tdb);
        if (rc != HVD_OK && rc != HVD_ERR_OVERFLOW) {
            free(tdb);
            return rc;
        }
        count += c;
    }
    free(tdb);
    *out_count = count;
    return count > cap ? HVD_ERR_OVERFLOW : HVD_OK;
}

int hvd_cpu_allpairs_hamming256(const uint8_t* db, int64_t n, const int32_t* group, int max_dist, hvd_pair* out,
                                int64_t cap, int64_t* out_count, int num_threads) {
    return hvd_cpu_allpairs_hamming256_rows(db, n, group, 0, n, max_dist, out, cap, out_count, num_threads);
}

/* Count-only variant for the CPU baseline timing (no output traffic). */
int64_t hvd_cpu_allpairs_count(const uint8_t* db, int64_t n, int max_dist, int num_threads) {
    int64_t count = 0;
    hvd_pair dummy;
    hvd_cpu_allpairs_hamming256_rows(db, n, NULL, 0, n, max_dist, &dummy, 0, &count, num_threads);
    return count;
}

/* ------------------------------------------------------- vPDQ matching -- */

/* vpdq matchTwoHashBrute restated on raw 32-byte frame hashes: for each query
 * frame, does ANY target frame lie within max_dist (comparator <=); and the
 * symmetric count. The caller turns (q_hits, t_hits) into a percentage with a
 * named policy (the reduction used by hvdaccelerators 0.4.0 is unpinned).
 * Mirrors vpdq.matchHashBytes(a, b, tol) of db/vptree.py:31. */
int hvd_cpu_match_two(const uint8_t* a, int64_t na, const uint8_t* b, int64_t nb, int max_dist, int32_t* q_hits,
                      int32_t* t_hits) {
    if (na < 0 || nb < 0 || !q_hits || !t_hits) return HVD_ERR_ARG;
    int32_t q = 0, t = 0;
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < nb; ++j)
            if (hvd_cpu_hamming256(a + 32 * i, b + 32 * j) <= max_dist) {
                q++;
                break;
            }
    for (int64_t j = 0; j < nb; ++j)
        for (int64_t i = 0; i < na; ++i)
            if (hvd_cpu_hamming256(a + 32 * i, b + 32 * j) <= max_dist) {
                t++;
                break;
            }
    *q_hits = q;
    *t_hits = t;
    return HVD_OK;
}

/* All video pairs a<b with at least one frame hit; frames is the concatenation of
 * all videos' frame hashes, offsets[V+1] the CSR boundaries (in frames).
 * Output sorted by (a, b). This is the brute-force ground truth that the
 * reference's VP-tree search approximates (dedup.py:445-502, db/vptree.py:664-815). */
int hvd_cpu_vpdq_match_videos(const uint8_t* frames, const int64_t* offsets, int64_t V, int max_dist,
                              hvd_vmatch* out, int64_t cap, int64_t* out_count) {
    if (V < 0 || !offsets || !out_count) return HVD_ERR_ARG;
    int64_t count = 0;
    for (int64_t a = 0; a < V; ++a)
        for (int64_t b = a + 1; b < V; ++b) {
            int32_t q, t;
            hvd_cpu_match_two(frames + 32 * offsets[a], offsets[a + 1] - offsets[a], frames + 32 * offsets[b],
                              offsets[b + 1] - offsets[b], max_dist, &q, &t);
            if (q > 0 || t > 0) {
                if (count < cap) out[count] = (hvd_vmatch){(uint32_t)a, (uint32_t)b, (uint32_t)q, (uint32_t)t};
                count++;
            }
        }
    *out_count = count;
    return count > cap ? HVD_ERR_OVERFLOW : HVD_OK;
}

/* The same brute force on num_threads host threads (bench.py's cpu_baseline wants every quota core, like the
 * all-pairs scan): video a is handled by thread a % T (rows shrink with a, so dealing them round-robin balances the
 * triangle), each thread keeps its records in (a, b) order in a private buffer, and the buffers are merged by a
 * T-way walk over a. Every record comes from the single-thread statement above (hvd_cpu_match_two). */
typedef struct {
    const uint8_t* frames;
    const int64_t* offsets;
    int64_t V;
    int max_dist, t, T;
    hvd_vmatch* out;
    int64_t cap, count;
} vmatch_job;

static void* vmatch_worker(void* arg) {
    vmatch_job* jb = (vmatch_job*)arg;
    for (int64_t a = jb->t; a < jb->V; a += jb->T)
        for (int64_t b = a + 1; b < jb->V; ++b) {
            int32_t q, t;
            hvd_cpu_match_two(jb->frames + 32 * jb->offsets[a], jb->offsets[a + 1] - jb->offsets[a],
                              jb->frames + 32 * jb->offsets[b], jb->offsets[b + 1] - jb->offsets[b], jb->max_dist, &q, &t);
            if (q > 0 || t > 0) {
                if (jb->count == jb->cap) {
                    jb->cap = jb->cap ? 2 * jb->cap : 1024;
                    jb->out = (hvd_vmatch*)realloc(jb->out, (size_t)jb->cap * sizeof(hvd_vmatch));
                }
                jb->out[jb->count++] = (hvd_vmatch){(uint32_t)a, (uint32_t)b, (uint32_t)q, (uint32_t)t};
            }
        }
    return NULL;
}

int hvd_cpu_vpdq_match_videos_mt(const uint8_t* frames, const int64_t* offsets, int64_t V, int max_dist,
                                 hvd_vmatch* out, int64_t cap, int64_t* out_count, int num_threads) {
    if (V < 0 || !offsets || !out_count) return HVD_ERR_ARG;
    int T = num_threads < 1 ? 1 : (num_threads > 256 ? 256 : num_threads);
    if (T > V) T = V > 0 ? (int)V : 1;
    vmatch_job jobs[256];
    pthread_t th[256];
    for (int t = 0; t < T; ++t) {
        jobs[t] = (vmatch_job){frames, offsets, V, max_dist, t, T, NULL, 0, 0};
        if (T == 1)
            vmatch_worker(&jobs[t]);
        else
            pthread_create(&th[t], NULL, vmatch_worker, &jobs[t]);
    }
    for (int t = 0; t < T && T > 1; ++t) pthread_join(th[t], NULL);
    int64_t count = 0, pos[256] = {0};
    for (int64_t a = 0; a < V; ++a) { /* thread a % T holds row a's records next in its buffer */
        vmatch_job* jb = &jobs[a % T];
        int64_t* p = &pos[a % T];
        while (*p < jb->count && jb->out[*p].a == (uint32_t)a) {
            if (count < cap) out[count] = jb->out[*p];
            count++;
            (*p)++;
        }
    }
    for (int t = 0; t < T; ++t) free(jobs[t].out);
    *out_count = count;
    return count > cap ? HVD_ERR_OVERFLOW : HVD_OK;
}
