"""Brute-force duplicate search on the GPU: the semantics of the reference's VP-tree
search (dedup.py:445-502, db/vptree.py:22-31,664-815) without the tree.

The reference discovers a pair (A,B) iff ``calculate_distance(A,B) <= search_threshold``
with ``calculate_distance = fix_vpdq_similarity(matchHashBytes(a, b, 31))`` and
``search_threshold = fix_vpdq_similarity(threshold)``, i.e. iff
``int(sim(A,B)) >= int(threshold)``. The VP-tree only approximates that
set (vPDQ similarity is not a metric and the tree is built from unseeded random samples,
db/vptree.py:431-441); this module computes it exactly.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, vpdq
from ._lib import PAIR_DTYPE, VMATCH_DTYPE

DISTANCE_TOLERANCE = 31  # per-frame Hamming tolerance (vpdqpy/vpdqpy.py:53, db/vptree.py:31)
DEFAULT_VARIANT = 13  # all-pairs kernel the product uses (FP4-MFMA, 128-bit first stage, form chosen by a probe); DESIGN.md 4.1


def fix_vpdq_similarity(similarity: float) -> int:
    """Turn [100.0, 0.0] similarity to [1, 101] (db/vptree.py:22-25)."""
    return (100 - int(similarity)) + 1


def calculate_distance(phash_a: bytes, phash_b: bytes) -> int:
    """Distance between two perceptual hashes, from [1, 101] (db/vptree.py:29-31)."""
    return fix_vpdq_similarity(vpdq.matchHashBytes(phash_a, phash_b, DISTANCE_TOLERANCE))


def allpairs_hamming(db: np.ndarray, max_dist: int = DISTANCE_TOLERANCE, group: np.ndarray | None = None,
                     cap: int | None = None) -> np.ndarray:
    """All i<j with hamming(db[i], db[j]) <= max_dist (and group[i] != group[j] if given),
    as a PAIR_DTYPE array sorted by (i, j). db: uint8[n,32]."""
    db = np.ascontiguousarray(db, dtype=np.uint8).reshape(-1, 32)
    n = db.shape[0]
    if group is not None:
        group = np.ascontiguousarray(group, dtype=np.int32)
        if group.shape != (n,):
            raise ValueError("group must have one int32 per hash")
    lib = _lib.ensure()
    cap = max(1024, n // 4) if cap is None else int(cap)
    while True:
        out = np.zeros(max(cap, 1), dtype=PAIR_DTYPE)
        cnt = C.c_int64(0)
        rc = lib.hvd_allpairs_hamming256(db.ctypes.data if n else None, n,
                                         group.ctypes.data if group is not None else None, int(max_dist),
                                         out.ctypes.data, cap, C.byref(cnt))
        if rc == _lib.HVD_ERR_OVERFLOW:  # reported, never truncated: retry with the exact size
            cap = int(cnt.value)
            continue
        _lib.check(rc)
        return out[: cnt.value].copy()


def match_videos(frames: np.ndarray, offsets: np.ndarray, max_dist: int = DISTANCE_TOLERANCE,
                 cap: int | None = None) -> np.ndarray:
    """Every video pair a<b with at least one frame hit, with its vPDQ counters, as a
    VMATCH_DTYPE array sorted by (a, b). frames: uint8[sum,32]; offsets: int64[V+1] (CSR)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, 32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    V = offsets.size - 1
    if V < 0 or (V >= 0 and offsets[-1] != frames.shape[0]):
        raise ValueError("offsets[-1] must equal the number of frame hashes")
    if max_dist < 0:  # comparator "lt" at tolerance 0: nothing can match
        return np.zeros(0, dtype=VMATCH_DTYPE)
    lib = _lib.ensure()
    cap = max(1024, V) if cap is None else int(cap)
    while True:
        out = np.zeros(max(cap, 1), dtype=VMATCH_DTYPE)
        cnt = C.c_int64(0)
        rc = lib.hvd_vpdq_match_videos(frames.ctypes.data if frames.size else None, offsets.ctypes.data, V,
                                       int(max_dist), out.ctypes.data, cap, C.byref(cnt))
        if rc == _lib.HVD_ERR_OVERFLOW:
            cap = int(cnt.value)
            continue
        _lib.check(rc)
        return out[: cnt.value].copy()


def match_videos_cross(frames_q: np.ndarray, offsets_q: np.ndarray, frames_t: np.ndarray, offsets_t: np.ndarray,
                       ids_q: np.ndarray | None = None, ids_t: np.ndarray | None = None,
                       max_dist: int = DISTANCE_TOLERANCE, cap: int | None = None) -> np.ndarray:
    """Query videos x target videos (batch form of VpTreeManager.search_file, db/vptree.py:865-902):
    VMATCH_DTYPE records (a = query index, b = target index) with >= 1 frame hit, sorted by (a, b).
    ids_q/ids_t (int32 per video): equal ids are never compared (a query that is in the target set)."""
    frames_q = np.ascontiguousarray(frames_q, dtype=np.uint8).reshape(-1, 32)
    frames_t = np.ascontiguousarray(frames_t, dtype=np.uint8).reshape(-1, 32)
    offsets_q = np.ascontiguousarray(offsets_q, dtype=np.int64)
    offsets_t = np.ascontiguousarray(offsets_t, dtype=np.int64)
    VQ, VT = offsets_q.size - 1, offsets_t.size - 1
    if offsets_q[-1] != frames_q.shape[0] or offsets_t[-1] != frames_t.shape[0]:
        raise ValueError("offsets[-1] must equal the number of frame hashes")
    if (ids_q is None) != (ids_t is None):
        raise ValueError("pass both id arrays or neither")
    if ids_q is not None:
        ids_q = np.ascontiguousarray(ids_q, dtype=np.int32)
        ids_t = np.ascontiguousarray(ids_t, dtype=np.int32)
        if ids_q.shape != (VQ,) or ids_t.shape != (VT,):
            raise ValueError("one id per video")
    if max_dist < 0:
        return np.zeros(0, dtype=VMATCH_DTYPE)
    lib = _lib.ensure()
    cap = max(1024, VQ) if cap is None else int(cap)
    while True:
        out = np.zeros(max(cap, 1), dtype=VMATCH_DTYPE)
        cnt = C.c_int64(0)
        rc = lib.hvd_vpdq_match_videos_cross(
            frames_q.ctypes.data if frames_q.size else None, offsets_q.ctypes.data, VQ,
            ids_q.ctypes.data if ids_q is not None else None,
            frames_t.ctypes.data if frames_t.size else None, offsets_t.ctypes.data, VT,
            ids_t.ctypes.data if ids_t is not None else None, int(max_dist), out.ctypes.data, cap, C.byref(cnt))
        if rc == _lib.HVD_ERR_OVERFLOW:
            cap = int(cnt.value)
            continue
        _lib.check(rc)
        return out[: cnt.value].copy()


def similarity_of_records(records: np.ndarray, lengths: np.ndarray, policy: str | None = None) -> np.ndarray:
    """Per-record similarity in [0,100] under the match policy, taking the better of the two
    search directions (the reference finds {A,B} from A's search or from B's)."""
    policy = vpdq.MATCH_POLICY if policy is None else policy
    na = lengths[records["a"]].astype(np.float64)
    nb = lengths[records["b"]].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        qp = np.where(na > 0, records["q_hits"] * 100.0 / na, 0.0)
        tp = np.where(nb > 0, records["t_hits"] * 100.0 / nb, 0.0)
    if policy == "min":
        return np.minimum(qp, tp)
    if policy in ("max", "query", "target"):
        return np.maximum(qp, tp)  # query(A,B)=q%, query(B,A)=t%: either direction reports the pair
    raise ValueError(f"unknown match policy {policy!r}")


def similar_video_pairs(records: np.ndarray, lengths: np.ndarray, threshold: float = 50.0,
                        policy: str | None = None) -> np.ndarray:
    """The duplicate-pair set of dedup.py:445-502: rows (a, b) with int(sim) >= int(threshold)."""
    if int(threshold) < 1:
        raise ValueError("threshold < 1 would select every pair of videos")
    sim = similarity_of_records(records, np.asarray(lengths), policy)
    keep = sim.astype(np.int64) >= int(threshold)  # int() truncation as in fix_vpdq_similarity
    return np.stack([records["a"][keep], records["b"][keep]], axis=1).astype(np.int64)


def find_potential_duplicates(video_hashes, threshold: float = 50.0, policy: str | None = None) -> list[tuple[int, int]]:
    """Counterpart of HydrusVideoDeduplicator.find_potential_duplicates (dedup.py:445-502)
    for an in-memory library: video_hashes is a sequence of VpdqHash / bytes; returns the
    sorted list of index pairs (a < b) that the reference would mark as potential
    duplicates (threshold default 50, entrypoint.py:55-57)."""
    blobs = [h.bytes if isinstance(h, vpdq.VpdqHash) else bytes(h) for h in video_hashes]
    for b in blobs:
        if len(b) % 32:
            raise ValueError("phash length not a multiple of 32")
    lengths = np.array([len(b) // 32 for b in blobs], dtype=np.int64)
    offsets = np.zeros(len(blobs) + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    frames = np.frombuffer(b"".join(blobs), dtype=np.uint8).reshape(-1, 32)
    recs = match_videos(frames, offsets, vpdq.frame_max_dist(DISTANCE_TOLERANCE))
    pairs = similar_video_pairs(recs, lengths, threshold, policy)
    return [(int(a), int(b)) for a, b in pairs]
