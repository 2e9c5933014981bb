"""Batch pipeline over a whole library: hash every video's frames in one pass over HBM, then
search all video pairs (BASELINE config 5 shape). The reference does this one video at a time
(`dedup.py:346-352`) and one tree probe per video (`dedup.py:468-475`); a GPU wants the batch.

Two forms:

* host arrays in, Python objects out (`hash_videos`, `dedupe_videos`) -- convenience, mirrors the
  per-video API;
* **device-resident** (`DeviceLibrary`, `dedupe_frames_on_device`): frames already in HBM are hashed,
  the quality filter (`VideoHasher.finish`, vpdqpy/vpdqpy.py:119: quality >= 31 kept,
  db/DedupeDB.py:550-553) runs as a stream compaction on the GPU together with the per-video CSR
  offsets, the kept hashes are rewritten as their FP4 image and searched, and the video-level
  vPDQ counters are reduced on the GPU. Nothing but `hvd_vmatch` records and the V+1 offsets
  crosses PCIe. With one process per GPU the frames are hashed in disjoint video ranges and the
  hash shards are all-gathered over RCCL before the sharded search.
"""

from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib, search, vpdq
from ._lib import VMATCH_DTYPE, DeviceBuffer


def hash_videos(videos) -> list[vpdq.VpdqHash]:
    """videos: sequence of uint8[n_i,h,w] / uint8[n_i,h,w,3] arrays with one common frame
    geometry. All frames go through the PDQ kernels in one batch; per video the frames with
    quality >= 31 are kept in order (VideoHasher.finish semantics, vpdqpy/vpdqpy.py:119)."""
    videos = [np.ascontiguousarray(v, dtype=np.uint8) for v in videos]
    if not videos:
        return []
    shape = videos[0].shape[1:]
    for v in videos:
        if v.shape[1:] != shape:
            raise ValueError("all videos must share one frame geometry")
    lengths = np.array([v.shape[0] for v in videos], dtype=np.int64)
    if lengths.sum() == 0:
        return [vpdq.VpdqHash(b"") for _ in videos]
    flat = np.concatenate([v for v in videos if v.shape[0]], axis=0)
    hashes, quality = vpdq.hash_frames(flat)
    out, pos = [], 0
    for n in lengths:
        h, q = hashes[pos : pos + n], quality[pos : pos + n]
        out.append(vpdq.VpdqHash(h[q >= vpdq.QUALITY_TOLERANCE].tobytes()))
        pos += n
    return out


def dedupe_videos(videos, threshold: float = 50.0, policy: str | None = None):
    """Frames in, duplicate pairs out: (phashes, pairs) with pairs = sorted index pairs (a<b) the
    reference would mark as potential duplicates at `threshold` (dedup.py:445-502)."""
    phashes = hash_videos(videos)
    return phashes, search.find_potential_duplicates(phashes, threshold, policy)


# ------------------------------------------------------------------ device-resident form ------


def hash_frames_on_device(d_frames_ptr: int, n: int, h: int, w: int, channels: int):
    """PDQ-hash n frames that already sit in HBM -> (d_hashes, d_quality) DeviceBuffers (n*32 B, int32[n]).
    Enqueued on the library stream; no host synchronisation."""
    lib = _lib.ensure()
    d_h = DeviceBuffer(32 * max(n, 1))
    d_q = DeviceBuffer(4 * max(n, 1))
    sb = C.c_size_t(0)
    _lib.check(lib.hvd_pdq_scratch_bytes(n, h, w, channels, C.byref(sb)))
    d_s = DeviceBuffer(sb.value) if sb.value else None
    _lib.check(lib.hvd_dev_pdq_hash_frames(d_frames_ptr, n, h, w, channels, d_s.ptr if d_s else None, d_h.ptr, d_q.ptr))
    if d_s is not None:
        _lib.check(lib.hvd_dev_sync())  # the scratch must outlive the kernels
        d_s.free()
    return d_h, d_q


class DeviceLibrary:
    """A video library resident in HBM: kept frame hashes (CSR by video), their FP4 image and the
    frame -> video map -- the operand of the video-level search."""

    def __init__(self, d_hashes: DeviceBuffer, d_offsets: DeviceBuffer, d_video: DeviceBuffer, n_frames: int,
                 n_videos: int):
        self.d_hashes, self.d_offsets, self.d_video = d_hashes, d_offsets, d_video
        self.n_frames, self.n_videos = int(n_frames), int(n_videos)
        self.d_img = None
        self._lengths = None

    @classmethod
    def from_raw_hashes(cls, d_hashes_ptr: int, d_quality_ptr: int, n: int, raw_offsets: np.ndarray,
                        min_quality: int = vpdq.QUALITY_TOLERANCE) -> "DeviceLibrary":
        """Quality filter + CSR on the device (hvd_dev_compact_kept). raw_offsets: int64[V+1] over the n
        hashed frames (host array; V+1 numbers are the only thing uploaded)."""
        lib = _lib.ensure()
        raw_offsets = np.ascontiguousarray(raw_offsets, dtype=np.int64)
        V = raw_offsets.size - 1
        if V < 0 or raw_offsets[0] != 0 or raw_offsets[-1] != n or (np.diff(raw_offsets) < 0).any():
            raise ValueError("raw_offsets must be a CSR over the n frames")
        d_roff = DeviceBuffer.from_array(raw_offsets)
        d_out_h = DeviceBuffer(32 * max(n, 1))
        d_out_off = DeviceBuffer(8 * (V + 1))
        d_out_vid = DeviceBuffer(4 * max(n, 1))
        kept = C.c_int64(0)
        try:
            _lib.check(lib.hvd_dev_compact_kept(d_hashes_ptr, d_quality_ptr, n, d_roff.ptr, V, int(min_quality),
                                                d_out_h.ptr, d_out_off.ptr, d_out_vid.ptr, C.byref(kept)))
        finally:
            d_roff.free()
        return cls(d_out_h, d_out_off, d_out_vid, kept.value, V)

    @classmethod
    def from_host(cls, frames: np.ndarray, offsets: np.ndarray) -> "DeviceLibrary":
        """Upload an existing library (hashes uint8[sum,32] + CSR offsets)."""
        lib = _lib.ensure()
        frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, 32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n, V = frames.shape[0], offsets.size - 1
        if V < 0 or offsets[-1] != n:
            raise ValueError("offsets[-1] must equal the number of frame hashes")
        d_h = DeviceBuffer.from_array(frames) if n else DeviceBuffer(1)
        d_off = DeviceBuffer.from_array(offsets)
        d_vid = DeviceBuffer(4 * max(n, 1))
        _lib.check(lib.hvd_dev_video_of_frames(d_off.ptr, V, n, d_vid.ptr))
        return cls(d_h, d_off, d_vid, n, V)

    def image(self) -> DeviceBuffer:
        """FP4 image of the kept hashes (built once, cached)."""
        if self.d_img is None:
            lib = _lib.ensure()
            sz = C.c_size_t(0)
            _lib.check(lib.hvd_fp4_image_bytes(self.n_frames, C.byref(sz)))
            self.d_img = DeviceBuffer(sz.value)
            _lib.check(lib.hvd_dev_expand_fp4(self.d_hashes.ptr, self.n_frames, self.d_img.ptr))
        return self.d_img

    def offsets(self) -> np.ndarray:
        return self.d_offsets.to_array(np.int64, self.n_videos + 1)

    def lengths(self) -> np.ndarray:
        if self._lengths is None:
            self._lengths = np.diff(self.offsets())
        return self._lengths

    def hashes(self) -> np.ndarray:
        return self.d_hashes.to_array(np.uint8, 32 * self.n_frames).reshape(-1, 32)

    def match_videos(self, max_dist: int | None = None, rank: int = 0, world: int = 1, cap: int | None = None) -> np.ndarray:
        """hvd_dev_vpdq_match_videos: VMATCH_DTYPE records sorted by (a, b); only these cross PCIe."""
        lib = _lib.ensure()
        max_dist = vpdq.frame_max_dist(search.DISTANCE_TOLERANCE) if max_dist is None else int(max_dist)
        if max_dist < 0 or self.n_frames < 2:
            return np.zeros(0, dtype=VMATCH_DTYPE)
        cap = max(4096, self.n_videos) if cap is None else int(cap)
        # the record buffer is kept from call to call (per context): two hipMalloc / hipFree round trips inside every search were
        # ~1 % of a 50 000-video pass during which the GPU did nothing. Buffer, launch and read-back stay under the context's
        # lock: two threads searching on one context would otherwise read each other's records, or free a buffer the other
        # is still reading back.
        with _record_lock():
            d_out, d_cnt, _ = _record_buffers(cap)  # (the call is told `cap` as asked, whatever the kept buffer would hold)
            _lib.check(lib.hvd_dev_vpdq_match_videos(self.image().ptr, self.n_frames, self.d_video.ptr, max_dist,
                                                     rank, world, d_out.ptr, cap, d_cnt.ptr))
            cnt = int(d_cnt.to_array(np.uint64, 1)[0])
            if cnt > cap:  # more video pairs than room: ONLY the emit is repeated (no second O(n^2) pass, no second
                # key exchange that every rank would have to enter in lock-step)
                cap = cnt
                d_out, d_cnt, _ = _record_buffers(cap)
                _lib.check(lib.hvd_dev_vpdq_emit_again(d_out.ptr, cap, d_cnt.ptr))
                cnt = int(d_cnt.to_array(np.uint64, 1)[0])
                assert cnt <= cap
            recs = d_out.to_array(VMATCH_DTYPE, cnt)
        # ((a, b) is unique, so the sort need not be stable: numpy's default 64-bit sort is three times faster on 21 k records)
        return recs[np.argsort((recs["a"].astype(np.uint64) << np.uint64(32)) | recs["b"])]

    def free(self) -> None:
        for b in (self.d_hashes, self.d_offsets, self.d_video, self.d_img):
            if b is not None:
                b.free()
        self.d_img = None


_RECORD_BUFFERS: dict = {}  # context index -> (d_out, d_cnt, cap): grow-only, released by release_record_buffers()
_RECORD_LOCKS: dict = {}    # context index -> lock held from the buffer lookup to the end of the read-back
_RECORD_MU = threading.Lock()


def _record_lock():
    ctx = _lib.load().hvd_get_context()
    with _RECORD_MU:
        lk = _RECORD_LOCKS.get(ctx)
        if lk is None:
            lk = _RECORD_LOCKS[ctx] = threading.Lock()
    return lk



def _record_buffers(cap: int):
    """(caller holds _record_lock())"""
    ctx = _lib.load().hvd_get_context()
    have = _RECORD_BUFFERS.get(ctx)
    if have is None or have[2] < cap or have[0].ptr is None:
        if have is not None:
            have[0].free()
            have[1].free()
        have = (DeviceBuffer(16 * cap), DeviceBuffer(8), cap)
        with _RECORD_MU:
            _RECORD_BUFFERS[ctx] = have
    return have


def release_record_buffers() -> None:
    with _RECORD_MU:
        for d_out, d_cnt, _ in _RECORD_BUFFERS.values():
            d_out.free()
            d_cnt.free()
        _RECORD_BUFFERS.clear()


def shard_frames(raw_offsets: np.ndarray, world: int) -> int:
    """Frames in the longest rank's share (shards are padded to it for the equal-size all-gather)."""
    V = raw_offsets.size - 1
    return max(int(raw_offsets[video_range_of_rank(V, r, world)[1]] - raw_offsets[video_range_of_rank(V, r, world)[0]])
               for r in range(world))


def gather_hash_shards(d_h: DeviceBuffer, d_q: DeviceBuffer, raw_offsets: np.ndarray, rank: int, world: int, exchange):
    """All-gather of the per-rank hash / quality shards (RCCL, equal-size padded blocks), then the padding is
    squeezed out so that the frames are in library order. -> (d_hashes, d_quality) for the whole library."""
    lib = _lib.ensure()
    V = raw_offsets.size - 1
    n_total = int(raw_offsets[-1])
    v_lo, v_hi = video_range_of_rank(V, rank, world)
    n_mine = int(raw_offsets[v_hi] - raw_offsets[v_lo])
    shard = max(1, shard_frames(raw_offsets, world))
    d_all_h, d_all_q = DeviceBuffer(32 * shard * world), DeviceBuffer(4 * shard * world)
    d_ph, d_pq = DeviceBuffer(32 * shard), DeviceBuffer(4 * shard)  # my shard, padded to the longest
    d_ph.zero()
    d_pq.zero()
    _lib.check(lib.hvd_memcpy_d2d(d_ph.ptr, d_h.ptr, 32 * n_mine))
    _lib.check(lib.hvd_memcpy_d2d(d_pq.ptr, d_q.ptr, 4 * n_mine))
    exchange.allgather_bytes_dev(d_ph.ptr, d_all_h.ptr, 32 * shard)
    exchange.allgather_bytes_dev(d_pq.ptr, d_all_q.ptr, 4 * shard)
    d_fh, d_fq = DeviceBuffer(32 * max(n_total, 1)), DeviceBuffer(4 * max(n_total, 1))
    for r in range(world):  # ranks own contiguous video ranges
        lo, hi = video_range_of_rank(V, r, world)
        a, b = int(raw_offsets[lo]), int(raw_offsets[hi])
        if b > a:
            _lib.check(lib.hvd_memcpy_d2d(d_fh.ptr + 32 * a, d_all_h.ptr + 32 * shard * r, 32 * (b - a)))
            _lib.check(lib.hvd_memcpy_d2d(d_fq.ptr + 4 * a, d_all_q.ptr + 4 * shard * r, 4 * (b - a)))
    _lib.check(lib.hvd_dev_sync())  # the staging buffers are freed below
    for buf in (d_all_h, d_all_q, d_ph, d_pq):
        buf.free()
    return d_fh, d_fq


def dedupe_frames_on_device(d_frames_ptr: int, raw_offsets: np.ndarray, h: int, w: int, channels: int,
                            threshold: float = 50.0, policy: str | None = None, rank: int = 0, world: int = 1,
                            exchange=None, keep_library: bool = False, timings: dict | None = None):
    """BASELINE config 5 for one rank: the frames of videos [v_lo, v_hi) of this rank sit at d_frames_ptr
    (world == 1: all of them); raw_offsets is the CSR of the WHOLE library. Hash -> (all-gather of the hash
    shards) -> quality filter + CSR -> FP4 image -> sharded video search with the counters reduced on the GPU
    -> pair predicate of dedup.py:445-502 on the few video-level records.
    -> (pairs int64[m,2], records, library or None). Every rank returns the same result.
    timings (optional dict): receives hash_ms and search_ms, HIP-event times on the library stream of the hash launch
    and of the whole video search (image, probe, all-pairs pass, key reduction, record emit); gather_ms (host clock: the
    all-gather of the hash shards over RCCL incl. the squeeze into library order, 0 at world 1), compact_ms (host clock:
    quality filter + CSR, synchronous) and the search's own phases from the library (host clock, hvd_debug_get
    vmatch_us_*): search_local_ms (packed hashes, probe, all-pairs pass, key set), search_exchange_ms (agreement words,
    all-gather of the key lists, merged set; 0 at world 1), search_fold_ms (keys -> pair map)."""
    import time
    raw_offsets = np.ascontiguousarray(raw_offsets, dtype=np.int64)
    V = raw_offsets.size - 1
    n_total = int(raw_offsets[-1])
    v_lo, v_hi = video_range_of_rank(V, rank, world)
    n_mine = int(raw_offsets[v_hi] - raw_offsets[v_lo])
    lib = _lib.ensure()

    def timed(key, fn):
        if timings is None:
            return fn()
        _lib.check(lib.hvd_timer_start())
        out = fn()
        ms = C.c_float(0)
        _lib.check(lib.hvd_timer_stop(C.byref(ms)))
        timings[key] = float(ms.value)
        return out

    d_h, d_q = timed("hash_ms", lambda: hash_frames_on_device(d_frames_ptr, n_mine, h, w, channels))
    t0 = time.perf_counter()
    if world > 1:
        if exchange is None:
            raise ValueError("world > 1 needs the RCCL exchange")
        d_fh, d_fq = gather_hash_shards(d_h, d_q, raw_offsets, rank, world, exchange)
        d_h.free()
        d_q.free()
        d_h, d_q = d_fh, d_fq
    t1 = time.perf_counter()
    library = DeviceLibrary.from_raw_hashes(d_h.ptr, d_q.ptr, n_total, raw_offsets)
    t2 = time.perf_counter()
    d_h.free()
    d_q.free()
    recs = timed("search_ms", lambda: library.match_videos(rank=rank, world=world))
    if timings is not None:
        timings["gather_ms"] = (t1 - t0) * 1e3 if world > 1 else 0.0
        timings["compact_ms"] = (t2 - t1) * 1e3
        us = C.c_int(0)
        for key in ("local", "exchange", "fold"):
            _lib.check(lib.hvd_debug_get(f"vmatch_us_{key}".encode(), C.byref(us)))
            timings[f"search_{key}_ms"] = us.value / 1e3
    pairs = search.similar_video_pairs(recs, library.lengths(), threshold, policy)
    if keep_library:
        return pairs, recs, library
    library.free()
    return pairs, recs, None


def dedupe_frames_in_process(frames_of_rank, raw_offsets: np.ndarray, h: int, w: int, channels: int,
                             threshold: float = 50.0, policy: str | None = None, timings: dict | None = None):
    """BASELINE config 5 on the library's in-process device group (hvd_init_devices / HVD_DEVICES), no launcher:
    one thread per context runs `dedupe_frames_on_device` as rank = context index. frames_of_rank(rank, world) ->
    device pointer of the frames of that rank's video range (`video_range_of_rank`), resident on that rank's device (it is
    called on the rank's thread, with the rank's context current). -> (pairs, records) -- every rank computes the same;
    rank 0's are returned. timings: rank 0's stage times."""
    from . import multigpu

    def one(rank, world):
        ex = multigpu.GroupExchange(rank, world) if world > 1 else None
        tm = {} if timings is not None and rank == 0 else None
        pairs, recs, _ = dedupe_frames_on_device(frames_of_rank(rank, world), raw_offsets, h, w, channels, threshold, policy,
                                                 rank, world, ex, timings=tm)
        if tm is not None:
            timings.update(tm)
        return pairs, recs

    results = multigpu.run_on_contexts(one)
    return results[0]


def video_range_of_rank(V: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous share of the videos hashed by `rank` (frames are independent: no collective while hashing)."""
    per = (V + world - 1) // world
    lo = min(V, rank * per)
    return lo, min(V, lo + per)
