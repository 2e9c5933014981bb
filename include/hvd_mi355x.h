/*
 * hvd_mi355x.h -- C-ABI of libhvd_mi355x.so: the MI355X (gfx950) replacement for the
 * native module `hvdaccelerators.vpdq` that hydrus-video-deduplicator calls for its
 * perceptual-hash hot path. Paths below are relative to the reference tree.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns
 * HVD_OK (0) or a negative error code, never throws, and never keeps a caller
 * pointer past the call. hvd_last_error() gives the message for the last failure on
 * the calling thread. There is NO CPU fallback: without a usable gfx950 device
 * hvd_init() fails with HVD_ERR_NO_DEVICE and every compute entry point fails with
 * HVD_ERR_STATE.
 *
 * Layouts
 *   frame hash  : 32 bytes = 256 bits; DCT coefficient bit k = i*16+j is byte k>>3,
 *                 bit k&7 (little-endian image of PDQ's uint16 w[16];
 *                 db/DedupeDB.py:535-559, dedup.py:83).
 *   video hash  : concatenation of N>=0 frame hashes (dedup.py:77-86).
 *   hvd_pair    : one frame-level hit (i<j, Hamming distance).
 *   hvd_vmatch  : one video-level hit (a<b) with the vPDQ counters: q_hits = frames
 *                 of a that have >=1 frame of b within max_dist, t_hits the converse.
 */
#ifndef HVD_MI355X_H
#define HVD_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVD_OK 0
#define HVD_ERR_ARG (-1)       /* bad argument */
#define HVD_ERR_HIP (-2)       /* HIP runtime error */
#define HVD_ERR_OVERFLOW (-3)  /* output buffer too small; *out_count holds the required size */
#define HVD_ERR_NO_DEVICE (-4) /* no gfx950 device visible */
#define HVD_ERR_RCCL (-5)      /* RCCL error */
#define HVD_ERR_STATE (-6)     /* hvd_init() not called / comm not initialised */

#define HVD_BYTES_PER_PDQ_HASH 32 /* == vpdq.VpdqHash.bytesPerPdqHash, dedup.py:83 */
#define HVD_UNIQUE_ID_BYTES 128
#define HVD_ABI_VERSION 6 /* 6 (round 6): + hvd_group_rearm; 5 (round 5): + hvd_hasher_acquire_n, hvd_hasher_commit_n, hvd_group_abort, hvd_runtime_info, hvd_timer_mark, hvd_timer_between; 4 (round 4): + hvd_init_devices, hvd_context_count, hvd_set_context, hvd_get_context, hvd_group_exchange; 3 (round 3): + hvd_host_malloc/free, hvd_hasher_set_threads, hvd_dev_vpdq_emit_again, hvd_comm_abort, hvd_dct_matrix_libm */
/* All-pairs kernel the host entry points use: FP4-MFMA with a 128-bit first stage; which of its two forms runs
 * (survivors fetch their other half | second stage out of registers) is chosen per launch from a probe of the data. */
#define HVD_DEFAULT_VARIANT 13

typedef struct {
    uint32_t i, j, dist, pad;
} hvd_pair;

typedef struct {
    uint32_t a, b, q_hits, t_hits;
} hvd_vmatch;

/* ------------------------------------------------------------ lifecycle -- */

int hvd_abi_version(void);
/* Number of visible HIP devices (0 and HVD_OK when there is none). */
int hvd_device_count(int* out_n);
/* Bind this process to one GPU, create the library stream, upload the 16x64 DCT matrix. Idempotent for the same
 * device. With HVD_DEVICES=0,1,2,... in the environment (the list must start with `device`) this is
 * hvd_init_devices() on that list: every binding of this library that calls hvd_init -- the vpdq-shaped Python
 * surface, the search, the VpTreeManager facade, the SQLite adapter -- then uses all listed GPUs, no launcher. */
int hvd_init(int device);
/* Bind this process to a GROUP of GPUs: one context per listed device (its own stream, scratch pool, select/context
 * words, communicator). The reference is ONE process (entrypoint.py:235 -> dedup.py:213); a group is how that one
 * process uses 8 MI355X. The host-buffer entry points (hvd_pdq_hash_frames_*, hvd_allpairs_hamming256,
 * hvd_vpdq_match_videos[_cross]) then shard by themselves -- frames in contiguous ranges; the hash DB replicated on
 * every device, tile (rb, cb) of the pair matrix on context (rb + cb) % n, one host thread per context, candidates /
 * video keys exchanged with RCCL all-gathers over xGMI (ncclCommInitAll: one communicator per device, one process) --
 * and return what a single device returns. Everything else works on the calling thread's CURRENT context
 * (hvd_set_context; context 0 by default): a caller that wants to drive the device-resident API on every GPU itself
 * runs one thread per context with rank = context index, world = group size (hvd_amd.pipeline, bench.py
 * --single-process). A device may be listed twice (two contexts, two streams on one GPU: a test configuration; RCCL
 * refuses duplicate devices, so such a group exchanges through host memory -- hvd_group_exchange() == 2). */
int hvd_init_devices(const int* devices, int n_devices);
int hvd_context_count(int* out_n);  /* contexts of the group (1 after hvd_init, 0 before) */
int hvd_set_context(int index);     /* the calling thread's current context (and HIP device) from now on */
int hvd_get_context(void);
int hvd_group_exchange(void);       /* 0: no group (one context); 1: RCCL between the devices; 2: host memory */
/* A caller that drives the contexts from its own threads (one per context) and fails on ONE of them before that thread
 * reaches an exchange step calls this so that the others do not wait for it for ever: the host-memory barrier is broken
 * (waiters return HVD_ERR_RCCL), RCCL communicators of the group are aborted (ncclCommAbort releases a collective that is
 * already waiting on the device). The group has no exchange afterwards until hvd_group_rearm() -- or hvd_init_devices()
 * with the same device list, which calls it -- forms it again; the library does the same by itself when one context of a
 * host-buffer call fails hard (a failure every rank left in lock-step through an agreement step aborts nothing). */
int hvd_group_abort(void);
/* ABI 6: put a group back to work after an abandoned exchange -- re-arms the host-memory barrier and re-creates RCCL
 * communicators that were aborted (ncclCommInitAll over the group's devices). Call it when no thread is inside a group call
 * (hvd_amd.multigpu.run_on_contexts does, before it starts its threads). HVD_OK on a healthy group; HVD_ERR_RCCL if the
 * communicators cannot be re-created (the group then exchanges through host memory). */
int hvd_group_rearm(void);
/* What this process runs on, as one JSON object in buf (NUL-terminated, truncated to len): HIP runtime / driver versions,
 * RCCL version and the path of the librccl that is actually loaded, every visible device (name, PCI bus id, gcnArch, CUs,
 * memory) and the peer matrix of the group's devices (hipDeviceCanAccessPeer, link type and hop count from
 * hipExtGetLinkTypeAndHopCount: xGMI or PCIe). bench.py prints it with every measurement so that a multi-GPU run can be
 * read without access to the box. Callable before hvd_init. */
int hvd_runtime_info(char* buf, size_t len);
int hvd_shutdown(void);
/* Copies the calling thread's last error message (NUL-terminated) into buf. */
int hvd_last_error(char* buf, size_t len);
/* The DCT matrix the kernels use (16*64 floats): the table compiled into the library (csrc/dct_table.inc), which is
 * authoritative -- hashes do not depend on the host's libm. hvd_dct_matrix_libm() recomputes it on this host the way
 * upstream does (float scale * double cos, rounded once); the parity tests assert the two are bit-identical. */
int hvd_dct_matrix(float* out_16x64);
int hvd_dct_matrix_libm(float* out_16x64);

/* ------------------------------------------ host-buffer entry points ------ */
/* These are what the Python `vpdq`-shaped shim binds; each stages through HBM,
 * runs the HIP kernels on the library stream and copies results back. */

/* Replaces the per-frame work of vpdq.VideoHasher.hash_frame (vpdqpy/vpdqpy.py:118)
 * for a batch of pre-decoded frames. frames: n*h*w bytes (gray; luma is defined as
 * the RGB formula with R=G=B) or n*h*w*3 bytes packed RGB24 row-major, exactly what
 * bytes(frame.planes[0]) yields at vpdqpy.py:118. h,w >= 64. Outputs: n*32 hash
 * bytes and n int32 qualities (0..100). Quality filtering (>=31 kept,
 * db/DedupeDB.py:550-553) is the caller's job (VideoHasher.finish, vpdqpy.py:119). */
int hvd_pdq_hash_frames_gray_u8(const uint8_t* frames, int64_t n, int h, int w, uint8_t* out_hashes,
                                int32_t* out_quality);
int hvd_pdq_hash_frames_rgb24_u8(const uint8_t* frames, int64_t n, int h, int w, uint8_t* out_hashes,
                                 int32_t* out_quality);

/* Replaces the O(visited nodes) stream of vpdq.matchHashBytes calls issued by the
 * VP-tree (db/vptree.py:29-31,737; dedup.py:445-502) with one brute-force pass:
 * all i<j with hamming(db[i],db[j]) <= max_dist and, when group != NULL,
 * group[i] != group[j]. out receives min(count,cap) records sorted by (i,j);
 * *out_count the true count (HVD_ERR_OVERFLOW if it exceeds cap). n < 2^32. */
int hvd_allpairs_hamming256(const uint8_t* db, int64_t n, const int32_t* group, int max_dist, hvd_pair* out,
                            int64_t cap, int64_t* out_count);

/* Replaces one vpdq.matchHash / vpdq.matchHashBytes call (vpdqpy/vpdqpy.py:56,
 * db/vptree.py:31): a, b are concatenated frame hashes (na, nb frames). Returns the
 * two vPDQ counters; the percentage policy lives in the host shim. */
int hvd_match_two(const uint8_t* a, int64_t na, const uint8_t* b, int64_t nb, int max_dist, int32_t* q_hits,
                  int32_t* t_hits);

/* Replaces HydrusVideoDeduplicator.find_potential_duplicates' tree search
 * (dedup.py:445-502) for a whole library: frames = all videos' frame hashes
 * concatenated, offsets[V+1] = CSR boundaries in frames. out receives every video
 * pair a<b with >=1 frame hit, sorted by (a,b). */
int hvd_vpdq_match_videos(const uint8_t* frames, const int64_t* offsets, int64_t V, int max_dist, hvd_vmatch* out,
                          int64_t cap, int64_t* out_count);

/* Query-set x target-set form of the same search: the steady-state workload after the first
 * run (new videos against the existing library; semantics of VpTreeManager.search_file,
 * db/vptree.py:865-902, for a batch of files). out: every (a = query video, b = target video)
 * with >=1 frame hit; q_hits counts frames of the query video, t_hits frames of the target.
 * ids_q / ids_t (both or neither): videos with equal ids are not compared with each other (a
 * query that is itself in the target set must not match itself). max_dist in [0,127]. */
int hvd_vpdq_match_videos_cross(const uint8_t* frames_q, const int64_t* offsets_q, int64_t VQ, const int32_t* ids_q,
                                const uint8_t* frames_t, const int64_t* offsets_t, int64_t VT, const int32_t* ids_t,
                                int max_dist, hvd_vmatch* out, int64_t cap, int64_t* out_count);

/* ------------------------------------------------ streaming frame hasher -- */
/* The native side of vpdq.VideoHasher (vpdqpy/vpdqpy.py:113-119): frames are pushed one at a
 * time (as a decoder yields them), staged in a ring of pinned batch slots, and each batch is
 * uploaded, hashed and downloaded on its own HIP stream, so PCIe transfer overlaps the kernels.
 * hvd_hasher_push blocks only when every slot is still in flight (back-pressure,
 * vpdqpy.py:115-117). One hasher per decoder thread; not thread-safe per handle. */
typedef struct hvd_hasher hvd_hasher;
int hvd_hasher_create(int width, int height, int channels, int64_t batch_frames, hvd_hasher** out);
int hvd_hasher_push(hvd_hasher* hs, const uint8_t* frame);
/* Host threads that share the copy of one frame inside hvd_hasher_push (the caller included; <= 0: library default = a quarter of the usable CPUs, between 2 and 8;
 * at most 8): what VideoHasher's num_threads (vpdqpy/vpdqpy.py:113) means on this path. One thread moves ~20 GB/s into
 * the pinned ring, the PCIe link behind it takes ~57. Frames below ~200 KB are copied by the caller alone. */
int hvd_hasher_set_threads(hvd_hasher* hs, int n);
/* Zero-copy feed: *out_frame is where the next frame (width*height*channels bytes) belongs inside the pinned
 * batch slot, so a decoder can reformat straight into it instead of handing over a copy (the reference copies
 * every frame into a Python bytes object, vpdqpy/vpdqpy.py:118); hvd_hasher_commit() makes it count.
 * acquire blocks like push; push == acquire + memcpy + commit. */
int hvd_hasher_acquire(hvd_hasher* hs, uint8_t** out_frame);
int hvd_hasher_commit(hvd_hasher* hs);
/* The same for a run of frames: *out_frames is where the next frames belong, back to back, *out_n (1 <= *out_n <= want)
 * how many fit there (what is left of the current batch: batches start small and grow per video); hvd_hasher_commit_n(n) makes the first n count
 * (0 <= n <= *out_n). One FFI round trip per run instead of two per frame: what a decoder of small frames wants. */
int hvd_hasher_acquire_n(hvd_hasher* hs, int64_t want, uint8_t** out_frames, int64_t* out_n);
int hvd_hasher_commit_n(hvd_hasher* hs, int64_t n);
int hvd_hasher_pending(hvd_hasher* hs, int64_t* out_frames);
/* All hashes (n*32 bytes) and qualities in push order; the hasher is reusable afterwards. */
int hvd_hasher_finish(hvd_hasher* hs, uint8_t* out_hashes, int32_t* out_quality, int64_t cap, int64_t* out_n);
int hvd_hasher_destroy(hvd_hasher* hs);

/* --------------------------------------------- device-resident API ------- */
/* For pipelines that keep data in HBM (hash on the GPU, then search) and for the
 * benchmark. Pointers named d_* are device pointers from hvd_dev_malloc. Kernels are
 * enqueued on the library stream and return immediately; hvd_dev_sync() waits. */

int hvd_dev_malloc(void** out_ptr, size_t bytes);
int hvd_dev_free(void* d_ptr);
/* Page-locked host memory: hvd_memcpy_h2d/d2h from/to it run at the DMA rate (a decoder that cannot write into
 * the hasher's slots -- hvd_hasher_acquire -- should at least decode into this); bench.py's H2D probe uses it. */
int hvd_host_malloc(void** out_ptr, size_t bytes);
int hvd_host_free(void* h_ptr);
int hvd_dev_memset(void* d_ptr, int value, size_t bytes);
int hvd_memcpy_h2d(void* d_dst, const void* src, size_t bytes);
int hvd_memcpy_d2h(void* dst, const void* d_src, size_t bytes);
int hvd_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes); /* enqueued on the library stream */
int hvd_dev_sync(void);
/* hipDeviceSynchronize(): every stream of the bound device (the library stream, the hashers' streams, RCCL's). */
int hvd_device_synchronize(void);

/* DCT accumulation mode of the frame hash. HVD_DCT_STRICT (default): every `sum += D*A` is a
 * separately rounded multiply and add -- the numerics of upstream's x86-64 builds and of the
 * oracle's default mode. HVD_DCT_FMA (opt-in): one fused multiply-add per step, executed by
 * v_mfma_f32_16x16x4_f32 -- the numerics upstream has where the compiler contracts that statement
 * (its arm64 builds); bit-exact against the oracle's fma mode, about 2 % of hash bits differ from
 * the strict mode. Quality values do not depend on the mode. Process-wide. */
#define HVD_DCT_STRICT 0
#define HVD_DCT_FMA 1
int hvd_set_pdq_dct_mode(int mode);
int hvd_get_pdq_dct_mode(void);

/* Developer switches for A/B measurements; results never change, only which kernel form runs:
 *   "pdq_dct_from_lds" 0|1|2|3 (SGPR | LDS | literals | by batch size), "pdq_luma_lut" 0|1|2   (64x64 hash kernel)
 *   "pdq_fused_down512" 0|1                                (0: generic 4-launch down-sampler)
 *   "pdq_down512_wave" 0|1|2                               (wave-per-frame kernel: never | batches >= 704 | always)
 *   "pdq_down512_wave_grid" n                              (waves in flight; 0 = what is resident at once)
 *   "pdq_down512_strip" 0|32|64                            (workgroup-per-frame kernel's strip width: by batch size | 32 | 64)
 *   "mfma_col_chunk_max"                                   (FP4-MFMA Hamming kernel: largest column chunk of a tile)
 *   "mfma_auto_mid" 0|18, "mfma_auto_mid_max_x100" n       (auto variant: may it pick the panel-mark queue form -- 18 -- and the
 *                                                           survivor density per 1024-pair tile, x 0.01, up to which it does -- 500)
 *   "mfma_queue_packed" 0|1                                (0: the pair queue settles from the FP4 images only)
 *   "mfma_force_sel" -1|0|1|2                              (which 128 bits the first stage sees: the probe's choice | bits 0..127 |
 *                                                           128..255 | 0..63 + 192..255)
 *   "pdq_hash_grid" n, "pdq_hash_prefetch" 0|1             (64x64 hash kernel: forced grid; next frame fetched ahead, off)
 *   "vmatch_exchange" 0|1|2                                (key exchange of the video search: iff world > 1 | always | never)
 *   "vmatch_slots_log2" 0|4..30                            (initial size of the video-reduction tables; tests the regrowth)
 *   "vmatch_variant" 0|8|9|12|13|18                        (all-pairs form of the video-level searches; 0 = the auto variant)
 *   "vmatch_bit_order" 0|1|2                               (video search: hashes rewritten with the 128 least entangled bits first: never |
 *                                                           from 65 536 frames on (default) | always; results never change)
 *   "match_server" 0|1                                     (hvd_match_two, small operands: one launch per call | a workgroup that stays
 *                                                           resident between calls and polls pinned host memory -- the default)
 *   "copy_nt" 0|1                                          (hvd_hasher_push: plain memcpy | non-temporal stores where the CPU has them)
 *   "mfma_clock_reset" 1                                   (telemetry: clear this context's clock accumulators, in stream order)
 * Unknown keys and out-of-range values return HVD_ERR_ARG. */
int hvd_debug_set(const char* key, int value);
/* "mfma_auto_form": the form (9, 18 or 12) the last auto-variant launch ran;
 * "mfma_probe_survivors" / "mfma_probe_survivors_hi" / "mfma_probe_survivors_mix": what its probe counted over bits 0..127 /
 * 128..255 / 0..63 + 192..255; "mfma_auto_half": the selection the first stage ran on (0 / 1 / 2 in that order).
 * Synchronises the library stream.
 * "vmatch_us_local" / "vmatch_us_exchange" / "vmatch_us_fold": host microseconds of the three phases of the last video-level
 * search on the calling thread's context (local: packed hashes, probe, all-pairs pass, key set; exchange: agreement words,
 * all-gather of the key lists, merged set -- 0 at world 1; fold: keys -> pair map). "copy_nt": 0 | 2 | 3 = plain memcpy |
 * AVX2 | AVX-512 streaming stores in hvd_hasher_push (hvd_debug_set "copy_nt" 0|1; HVD_COPY_NT=0 in the environment).
 * "vmatch_bit_order_used": 1 if the last video search on this context rewrote its hashes in a chosen bit order.
 * "hasher_us_copy" / "hasher_us_submit" / "hasher_us_wait": host microseconds the streaming hashers of this process spent copying
 * frames into the ring, enqueueing batches and waiting for a slot since the last read (reading clears).
 * "mfma_pass_khz": the shader clock (kHz) the FP4-MFMA all-pairs passes of this context ran at since the last
 * "mfma_clock_reset": one workgroup in eight brackets its lifetime with s_memtime (shader cycles) and s_memrealtime (constant
 * rate); cycles / ticks x hipDeviceAttributeWallClockRate. "mfma_clock_samples": how many workgroups contributed (0: none,
 * and "mfma_pass_khz" reads 0). Both wait for the library stream. */
int hvd_debug_get(const char* key, int* out_value);

/* Bytes of device scratch hvd_dev_pdq_hash_frames needs for this geometry (0 for
 * 64x64 gray): the 64x64 float luma of every frame plus the blur workspace. */
int hvd_pdq_scratch_bytes(int64_t n, int h, int w, int channels, size_t* out_bytes);
/* channels: 1 (gray u8) or 3 (RGB24). d_scratch: hvd_pdq_scratch_bytes() bytes
 * (NULL when that is 0). h,w in [64,4096]. */
int hvd_dev_pdq_hash_frames(const void* d_frames, int64_t n, int h, int w, int channels, void* d_scratch,
                            void* d_hashes, void* d_quality);

/* Brute-force pass over the tiles owned by `rank` of `world` (tile (rb,cb) belongs
 * to rank (rb+cb) % world; world=1 => everything). Appends hvd_pair records to
 * d_pairs[cap] and bumps the uint64 at d_count (the caller zeroes it). Records are
 * unordered. variant: 0 = the integer form of the north star (8 xor + 8 popcount per comparison), 1 = the same behind a
 * 128-bit prefilter (the independent path of the full-size parity tests). */
int hvd_dev_allpairs_hamming256(const void* d_db, int64_t n, const void* d_group, int max_dist, int rank, int world,
                                void* d_pairs, int64_t cap, void* d_count, int variant);

/* Matrix-core form of the same pass (variants 8, 9, 12, 18 and 13 = chosen per launch by a probe): the DB is first
 * rewritten as its FP4 image (every bit b as the e2m1 number 1-2b; 128 bytes per hash,
 * rows padded to a multiple of 1024), then v_mfma_f32_32x32x64_f8f6f4 produces
 * 256 - 2*hamming for 32x32 pairs at a time. Output contract identical to
 * hvd_dev_allpairs_hamming256 (bit-identical pair set). */
int hvd_fp4_image_bytes(int64_t n, size_t* out_bytes);
int hvd_dev_expand_fp4(const void* d_db, int64_t n, void* d_img);
int hvd_dev_allpairs_hamming256_mfma(const void* d_db, const void* d_img, int64_t n, const void* d_group, int max_dist, int rank,
                                     int world, void* d_pairs, int64_t cap, void* d_count, int variant);

/* Rectangular form on two FP4 images (nq query hashes x nt target hashes): appends (i = query
 * row, j = target row, dist). d_group_q / d_group_t (both or neither): pairs with equal group
 * values are dropped. max_dist in [0,127]. */
int hvd_dev_cross_hamming256_mfma(const void* d_img_q, int64_t nq, const void* d_img_t, int64_t nt,
                                  const void* d_group_q, const void* d_group_t, int max_dist, int rank, int world,
                                  void* d_pairs, int64_t cap, void* d_count);

/* ---- video-level search with everything resident in HBM (BASELINE config 5: hash on the GPU, then search) ----
 * The three calls below use library-owned grow-only scratch, synchronise the library stream before returning
 * and are serialised against each other. */

/* d_out_video[f] = index of the video that owns frame f, from CSR offsets (int64[V+1] on the device). */
int hvd_dev_video_of_frames(const void* d_offsets, int64_t V, int64_t n, void* d_out_video);

/* VideoHasher.finish() for a whole library at once (vpdqpy/vpdqpy.py:119; keep/drop contract of dedup.py:74-86,
 * quality >= min_quality kept as in db/DedupeDB.py:550-553): stream compaction of the kept frame hashes in frame
 * order. In: d_hashes n*32 B, d_quality int32[n], d_offsets int64[V+1] over the n raw frames. Out: d_out_hashes
 * (room for n*32 B), d_out_offsets int64[V+1] over the kept frames, d_out_video int32[kept] (room for n),
 * *out_kept. A video may end up with 0 frames (legal: dedup.py:82-86). */
int hvd_dev_compact_kept(const void* d_hashes, const void* d_quality, int64_t n, const void* d_offsets, int64_t V,
                         int min_quality, void* d_out_hashes, void* d_out_offsets, void* d_out_video, int64_t* out_kept);

/* Every video pair a<b with >= 1 frame hit, with its vPDQ counters (semantics of vpdqpy/vpdqpy.py:49-56 for all
 * pairs at once; replaces the tree walk of dedup.py:468-475). d_img: FP4 image of the n frame hashes; d_video:
 * int32[n] frame -> video (frames of one video are never compared). The counters are reduced ON THE DEVICE: the
 * all-pairs kernel records "frame f has a match in video v" in a set in HBM, which is then folded into one
 * hvd_vmatch per video pair -- frame-level hits never leave the GPU. world > 1: this rank compares its tiles,
 * the key sets are all-gathered over RCCL (hvd_comm_init required) and every rank returns the full result.
 * d_out[cap] receives min(count, cap) unordered records, the uint64 at d_count the true count. max_dist in
 * [0,127]. */
int hvd_dev_vpdq_match_videos(const void* d_img, int64_t n, const void* d_video, int max_dist, int rank, int world,
                              void* d_out, int64_t cap, void* d_count);
/* The records of the LAST hvd_dev_vpdq_match_videos[_cross] call once more, into a (larger) buffer: only the emit
 * step runs -- no compare, no exchange, so in a multi-rank pass a rank whose buffer was too small does not drag the
 * others into another collective. The pair map of that call stays valid until the next video search on this
 * process (the host-buffer entry points hvd_vpdq_match_videos[_cross] included). */
int hvd_dev_vpdq_emit_again(void* d_out, int64_t cap, void* d_count);
/* Query library x target library form (VpTreeManager.search_file for a batch, db/vptree.py:865-902).
 * d_excl_q / d_excl_t (both or neither): int32 per frame, frames with equal values are not compared. */
int hvd_dev_vpdq_match_videos_cross(const void* d_img_q, int64_t nq, const void* d_video_q, const void* d_excl_q,
                                    const void* d_img_t, int64_t nt, const void* d_video_t, const void* d_excl_t,
                                    int max_dist, int rank, int world, void* d_out, int64_t cap, void* d_count);

/* Host-only: the tile geometry hvd_dev_allpairs_hamming256 uses for (n, variant): a
 * tile is rows [rb*rows_per_block, +rows_per_block) x columns [cb*col_chunk, +col_chunk). */
int hvd_allpairs_tile_geometry(int64_t n, int variant, uint32_t* rows_per_block, uint32_t* col_chunk);

/* hipEvent pair on the library stream: wall time of everything enqueued between. */
int hvd_timer_start(void);
int hvd_timer_stop(float* out_ms);
/* Eight event slots per context for split timings without extra synchronisation (ABI 5): hvd_timer_mark(k) records event k
 * on the library stream and returns at once; hvd_timer_between(a, b, &ms) waits for event b and returns the device time
 * between the two. bench.py marks expand | kernel | end of a step and reads both intervals after the step's own sync. */
int hvd_timer_mark(int slot);
int hvd_timer_between(int slot_a, int slot_b, float* out_ms);

/* ----------------------------------------------- multi-GPU exchange ------ */
/* One process per GPU; rank 0 creates the id, the caller's control channel hands it to the other ranks
 * (hvd_amd.rendezvous: a loopback TCP star -- no torch; bench.py and hvd_amd.multigpu.connect_rccl use it),
 * every rank calls hvd_comm_init (collective). */
int hvd_comm_unique_id(uint8_t out_id[HVD_UNIQUE_ID_BYTES]);
int hvd_comm_init(const uint8_t id[HVD_UNIQUE_ID_BYTES], int rank, int world);
/* RCCL all-gather over xGMI of each rank's candidate pairs: counts first, then the
 * records padded to the max count. Every rank receives the concatenation (rank
 * order) in out_host[cap]; *out_total is the total number of records. */
int hvd_comm_allgather_pairs(const void* d_pairs, int64_t count, hvd_pair* out_host, int64_t cap, int64_t* out_total);
/* RCCL all-gather of equally sized device buffers (hash shards produced on-device). */
int hvd_comm_allgather_bytes(const void* d_send, void* d_recv, size_t bytes_per_rank);
int hvd_comm_destroy(void);
/* Tear the communicator down WITHOUT the collective handshake of ncclCommDestroy: for a rank whose own
 * hvd_comm_init succeeded while another rank's failed or timed out (hvd_amd.multigpu.connect_rccl) -- the
 * half-formed communicator must neither be used nor destroyed normally. Idempotent. */
int hvd_comm_abort(void);

#ifdef __cplusplus
}
#endif
#endif /* HVD_MI355X_H */
