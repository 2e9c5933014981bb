// hvd_kernels.h -- internal launch interface between the C-ABI host layer
// (hvd_api.cpp) and the gfx950 kernels (k_hamming.hip, k_pdq.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hvd_mi355x.h"
#include "hvd_devhash.h"

namespace hvd {

struct AllPairsArgs {
    const void* d_db;        // n * 32 bytes (FP4-MFMA forms: optional, the packed hashes of the image's rows)
    const void* d_db_q = nullptr;  // rectangular form: the packed query hashes (optional)
    uint32_t n;
    const int32_t* d_group;  // nullable
    uint32_t max_dist;
    uint32_t rank, world;
    hvd_pair* d_pairs;
    unsigned long long cap;
    unsigned long long* d_count;
    int variant;
    uint32_t col_chunk;      // 0 = pick automatically
    int ctx_id = 0;          // the caller's context: selects the FP4-MFMA forms' per-context select/context words
    VideoSink sink = {nullptr, 0, nullptr, nullptr, nullptr};  // FP4-MFMA form only: reduce to video level (K3)
    // auto variant only. false: probe and all candidate forms are enqueued, the unchosen ones return at once -- no host
    // synchronisation (the device-resident entry points promise that). true (callers that wait for the result anyway: the video
    // search, the host-buffer entry): the host reads the probe's decision and launches the chosen form alone -- on a 2.9 M-frame
    // library the two empty launches are ~1e6 workgroups each, 0.2 + 0.4 ms.
    bool sync_decide = false;
};

hipError_t launch_allpairs(const AllPairsArgs& a, hipStream_t s);
bool allpairs_geometry(uint32_t n, int variant, uint32_t* rows_per_block, uint32_t* col_chunk);

// FP4-MFMA forms (k_hamming_mfma.hip), variants 8, 9, 12, 13 (auto), 18. d_img: fp4_rows_padded(n)*128 bytes.
uint32_t fp4_rows_padded(uint32_t n);
extern uint32_t g_mfma_col_chunk_max;
extern uint32_t g_mfma_auto_mid, g_mfma_auto_mid_max_x100, g_mfma_queue_packed;
extern int g_mfma_force_sel;
hipError_t launch_expand_fp4(const void* d_db, uint32_t n, void* d_img, hipStream_t s);
hipError_t launch_pack_fp4(const void* d_img, uint32_t n, void* d_db, hipStream_t s);  // image -> packed 32-byte hashes
// data-dependent bit order of the video search (k_hamming_mfma.hip): co-occurrence counts of a strided sample, and the rewrite
hipError_t launch_bit_cooc(const void* d_bits, uint32_t stride, uint32_t words, void* d_rows, void* d_cooc, hipStream_t s);
hipError_t launch_reorder_bits(const void* d_bits_in, uint32_t n, const uint8_t perm[256], void* d_bits_out, void* d_img, hipStream_t s);
hipError_t launch_allpairs_mfma(const AllPairsArgs& a, const void* d_img, hipStream_t s);
// a.n / a.d_group describe the target set; rows are the nq query hashes (image d_img_q, groups a.d_group
// for queries and d_group_t for targets; pass both or neither).
hipError_t launch_cross_mfma(const AllPairsArgs& a, const void* d_img_q, uint32_t nq, const void* d_img_t,
                             const int32_t* d_group_t, hipStream_t s);
hipError_t mfma_select_buffer(int ctx_id, uint32_t** out);  // per context; [0] = form the auto variant ran last, [1] = probe survivors
// clock telemetry of a context's all-pairs passes: {shader cycles, constant-rate ticks, sampled workgroups, 0} since the reset
hipError_t mfma_clock_reset(int ctx_id, hipStream_t s);
hipError_t mfma_clock_read(int ctx_id, hipStream_t s, unsigned long long out[4]);
void mfma_release();
void pdq_release();           // k_pdq.hip: free the hash kernel's work-counter ring (hvd_shutdown)
void stream_release_cache();  // hvd_stream.cpp: free the parked hasher slot sets (hvd_shutdown)
void stream_set_copy_nt(int on);  // hvd_stream.cpp / copy_pool.h: non-temporal stores into the pinned ring (debug key "copy_nt")
int stream_copy_nt_level();
long long stream_take_ns(int which);  // 0 copy, 1 submit, 2 wait: host time of the streaming feed since the last read
bool allpairs_mfma_geometry(uint32_t n, int variant, uint32_t* rows_per_block, uint32_t* col_chunk);

// Video-level reduction and quality compaction (k_vmatch.hip).
hipError_t launch_keys_to_pairs(const unsigned long long* d_src, unsigned long long n_src, const int32_t* d_vid_q,
                                const int32_t* d_vid_t, bool rect, unsigned long long* d_pkeys, void* d_pcnt,
                                unsigned long long pmask, unsigned long long* d_counters, hipStream_t s);
hipError_t launch_pairs_emit(const unsigned long long* d_pkeys, const void* d_pcnt, unsigned long long slots, hvd_vmatch* d_out,
                             unsigned long long cap, unsigned long long* d_count, hipStream_t s);
hipError_t launch_set_to_list(const unsigned long long* d_tab, unsigned long long slots, unsigned long long* d_list,
                              unsigned long long cap, unsigned long long* d_count, hipStream_t s);
hipError_t launch_list_to_set(const unsigned long long* d_list, unsigned long long n, unsigned long long* d_tab,
                              unsigned long long mask, unsigned long long* d_counters, hipStream_t s);
size_t compact_scratch_bytes(unsigned long long n);
hipError_t launch_compact_kept(const void* d_hashes, const int32_t* d_quality, unsigned long long n, const long long* d_offsets,
                               uint32_t V, int min_q, void* d_out_hashes, long long* d_out_offsets, int32_t* d_out_video,
                               void* d_scratch, unsigned long long* d_total, hipStream_t s);
hipError_t launch_video_of_frames(const long long* d_offsets, uint32_t V, unsigned long long n, int32_t* d_out_video,
                                  hipStream_t s);

// Synthetic 64x64 gray video frames generated in HBM (k_synth.hip; workload generator, not on the hashing path).
hipError_t launch_synth_frames64(uint8_t* d_out, long long v0, uint32_t frames_per_video, unsigned long long n_frames,
                                 uint64_t seed, const int32_t* d_copy_of, hipStream_t s);

hipError_t launch_match_two(const uint32_t* d_a, uint32_t na, const uint32_t* d_b, uint32_t nb, uint32_t max_dist,
                            uint32_t* d_tflags, int32_t* d_hits, hipStream_t s);
uint32_t match_two_small_limit();
hipError_t launch_match_server(const uint32_t* ops, int32_t* hdr, uint32_t last, int32_t launch_id, unsigned long long idle_ticks,
                               unsigned long long life_ticks, hipStream_t s);
hipError_t launch_match_two_small(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb, uint32_t max_dist,
                                  int32_t* hits, int32_t seq, hipStream_t s);

// PDQ frame hashing. d_dct: 16*64 floats (host-computed, uploaded once).
// kind 0: gray u8 64x64 frames; kind 1: float 64x64 buffers (output of the
// down-sampler). d_in strides are implied by kind.
extern int g_pdq_dct_from_lds;
void pdq_dct_table_copy(float* out_16x64);               // the compiled-in DCT matrix (csrc/dct_table.inc): authoritative
bool pdq_dct_table_matches(const float* host_16x64);  // the kernels' compile-time DCT table vs the host's computation
extern int g_pdq_luma_lut;
extern int g_pdq_hash_grid;
extern int g_pdq_hash_prefetch;
extern int g_pdq_dct_mode;
extern bool g_pdq_fused_down512;
extern int g_pdq_down512_wave;
extern int g_pdq_down512_wave_grid;
extern int g_pdq_down512_strip;
hipError_t launch_pdq_hash64(const void* d_in, int kind, int64_t n, const float* d_dct, uint8_t* d_hashes,
                             int32_t* d_quality, hipStream_t s);

// Luma + 2x Jarosz box filter + decimate to 64x64 float, for h,w != 64 (the
// reference's 512x512 rgb24 frames, vpdqpy/vpdqpy.py:90-95,113).
// d_ws: min(n,1024) * pdq_downsample_ws_floats(h,w) floats of workspace.
size_t pdq_downsample_ws_floats(int h, int w);
hipError_t launch_pdq_downsample(const uint8_t* d_frames, int64_t n, int h, int w, int channels, float* d_ws,
                                 float* d_out64, hipStream_t s);
// 64x64 rgb24 frames: luma only (no blur, as upstream's 64x64 shortcut).
hipError_t launch_pdq_luma64_rgb(const uint8_t* d_frames, int64_t n, float* d_out64, hipStream_t s);

}  // namespace hvd
