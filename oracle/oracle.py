"""ctypes loader for the CPU oracle (oracle/libhvd_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The product package never imports this module.
PARITY UNPINNED -- see the header of oracle/hvd_oracle.c.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhvd_oracle.so")

PAIR_DTYPE = np.dtype([("i", "<u4"), ("j", "<u4"), ("dist", "<u4"), ("pad", "<u4")])
VMATCH_DTYPE = np.dtype([("a", "<u4"), ("b", "<u4"), ("q_hits", "<u4"), ("t_hits", "<u4")])


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "hvd_oracle.c")
    if force or not os.path.exists(_SO) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libhvd_oracle.so"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, i32p, i64p, f32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_float))
        L.hvd_cpu_dct_matrix.restype = f32p
        for name in ("hvd_cpu_pdq_hash_frames_gray_u8", "hvd_cpu_pdq_hash_frames_rgb24_u8"):
            fn = getattr(L, name)
            fn.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
            fn.restype = C.c_int
        L.hvd_cpu_set_dct_mode.argtypes = [C.c_int]
        L.hvd_cpu_set_dct_mode.restype = None
        L.hvd_cpu_hamming256.argtypes = [C.c_void_p, C.c_void_p]
        L.hvd_cpu_hamming256.restype = C.c_int
        L.hvd_cpu_allpairs_hamming256_rows.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                                       C.c_int, C.c_void_p, C.c_int64, i64p, C.c_int]
        L.hvd_cpu_allpairs_hamming256_rows.restype = C.c_int
        L.hvd_cpu_allpairs_hamming256_bands.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                                        C.c_void_p, C.c_int64, i64p, C.c_int]
        L.hvd_cpu_allpairs_hamming256_bands.restype = C.c_int
        L.hvd_cpu_allpairs_count.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.hvd_cpu_allpairs_count.restype = C.c_int64
        L.hvd_cpu_match_two.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, i32p, i32p]
        L.hvd_cpu_match_two.restype = C.c_int
        L.hvd_cpu_vpdq_match_videos.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                                i64p]
        L.hvd_cpu_vpdq_match_videos.restype = C.c_int
        L.hvd_cpu_vpdq_match_videos_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                                   i64p, C.c_int]
        L.hvd_cpu_vpdq_match_videos_mt.restype = C.c_int
        _lib = L
    return _lib


def dct_matrix() -> np.ndarray:
    p = lib().hvd_cpu_dct_matrix()
    return np.ctypeslib.as_array(p, shape=(16, 64)).copy()


def hash_frames(frames: np.ndarray, num_threads: int = 1, want_coeffs: bool = False, fma: bool = False):
    """frames: uint8[n,h,w] (gray) or uint8[n,h,w,3] (rgb24) -> (hashes u8[n,32], quality i32[n][, coeffs f32[n,256]]).
    fma: DCT accumulation by fused multiply-add (upstream's arm64 numerics) instead of mul-then-add."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    if frames.ndim == 3:
        fn = lib().hvd_cpu_pdq_hash_frames_gray_u8
    elif frames.ndim == 4 and frames.shape[3] == 3:
        fn = lib().hvd_cpu_pdq_hash_frames_rgb24_u8
    else:
        raise ValueError("frames must be uint8[n,h,w] or uint8[n,h,w,3]")
    n, h, w = frames.shape[:3]
    hashes = np.zeros((n, 32), dtype=np.uint8)
    quality = np.zeros(n, dtype=np.int32)
    coeffs = np.zeros((n, 256), dtype=np.float32) if want_coeffs else None
    lib().hvd_cpu_set_dct_mode(1 if fma else 0)
    try:
        rc = fn(frames.ctypes.data, n, h, w, hashes.ctypes.data, quality.ctypes.data,
                coeffs.ctypes.data if want_coeffs else None, num_threads)
    finally:
        lib().hvd_cpu_set_dct_mode(0)
    if rc != 0:
        raise RuntimeError(f"oracle hash_frames rc={rc}")
    return (hashes, quality, coeffs) if want_coeffs else (hashes, quality)


def hamming256(a, b) -> int:
    a = np.frombuffer(bytes(a), dtype=np.uint8)
    b = np.frombuffer(bytes(b), dtype=np.uint8)
    assert a.size == 32 and b.size == 32
    return lib().hvd_cpu_hamming256(a.ctypes.data, b.ctypes.data)


def allpairs(db: np.ndarray, max_dist: int = 31, group: np.ndarray | None = None, rows: tuple[int, int] | None = None,
             cap: int = 1 << 20, num_threads: int = 1) -> np.ndarray:
    """All i<j with hamming <= max_dist, sorted by (i,j); structured array PAIR_DTYPE."""
    db = np.ascontiguousarray(db, dtype=np.uint8).reshape(-1, 32)
    n = db.shape[0]
    if group is not None:
        group = np.ascontiguousarray(group, dtype=np.int32)
        assert group.shape == (n,)
    r0, r1 = rows if rows is not None else (0, n)
    out = np.zeros(max(cap, 1), dtype=PAIR_DTYPE)
    cnt = C.c_int64(0)
    rc = lib().hvd_cpu_allpairs_hamming256_rows(db.ctypes.data, n, group.ctypes.data if group is not None else None,
                                                r0, r1, max_dist, out.ctypes.data, cap, C.byref(cnt), num_threads)
    if rc == -3:
        return allpairs(db, max_dist, group, rows, cap=int(cnt.value), num_threads=num_threads)
    if rc != 0:
        raise RuntimeError(f"oracle allpairs rc={rc}")
    res = out[: cnt.value].copy()
    res.sort(order=["i", "j"])
    return res


def allpairs_bands(db: np.ndarray, bands, max_dist: int = 31, group: np.ndarray | None = None, cap: int = 1 << 20,
                   num_threads: int = 1) -> np.ndarray:
    """The brute force restricted to the rows of `bands` = [(row_begin, row_end), ...] (ascending, disjoint): all (i, j)
    with i in a band, i < j < n, hamming <= max_dist; sorted by (i, j). One block-transposed copy of the DB serves every
    band (the full-size checks of DBs too large to scan whole: BASELINE configs[3])."""
    db = np.ascontiguousarray(db, dtype=np.uint8).reshape(-1, 32)
    n = db.shape[0]
    b = np.ascontiguousarray(np.asarray(bands, dtype=np.int64).reshape(-1, 2))
    if group is not None:
        group = np.ascontiguousarray(group, dtype=np.int32)
        assert group.shape == (n,)
    out = np.zeros(max(cap, 1), dtype=PAIR_DTYPE)
    cnt = C.c_int64(0)
    rc = lib().hvd_cpu_allpairs_hamming256_bands(db.ctypes.data, n, group.ctypes.data if group is not None else None,
                                                 b.ctypes.data, b.shape[0], max_dist, out.ctypes.data, cap, C.byref(cnt),
                                                 num_threads)
    if rc == -3:
        return allpairs_bands(db, b, max_dist, group, cap=int(cnt.value), num_threads=num_threads)
    if rc != 0:
        raise RuntimeError(f"oracle allpairs_bands rc={rc}")
    return out[: cnt.value].copy()


def allpairs_count(db: np.ndarray, max_dist: int = 31, num_threads: int = 1, native: bool = False) -> int:
    db = np.ascontiguousarray(db, dtype=np.uint8).reshape(-1, 32)
    L = native_lib() if native else lib()
    return int(L.hvd_cpu_allpairs_count(db.ctypes.data, db.shape[0], max_dist, num_threads))


NATIVE_FLAGS = "-O3 -march=native"
PORTABLE_FLAGS = "-O2 -mpopcnt"
_native = None


def native_lib():
    """The oracle rebuilt on THIS host with -O3 -march=native (cpu_baseline leg of bench.py; `make native`), or the
    portable library when no compiler is available here. Only the all-pairs counter is bound."""
    global _native
    if _native is None:
        so = os.path.join(_HERE, "libhvd_oracle_native.so")
        try:
            subprocess.check_call(["make", "-C", _HERE, "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            L = C.CDLL(so)
            L.flags = NATIVE_FLAGS
        except (OSError, subprocess.CalledProcessError):
            L = C.CDLL(build())
            L.flags = PORTABLE_FLAGS + " (no compiler on this host: portable build)"
        L.hvd_cpu_allpairs_count.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.hvd_cpu_allpairs_count.restype = C.c_int64
        L.hvd_cpu_allpairs_uses_avx512.restype = C.c_int
        _native = L
    return _native


def uses_avx512(native: bool = False) -> bool:
    L = native_lib() if native else lib()
    L.hvd_cpu_allpairs_uses_avx512.restype = C.c_int
    return bool(L.hvd_cpu_allpairs_uses_avx512())


def match_two(a: bytes, b: bytes, max_dist: int = 31) -> tuple[int, int]:
    """(q_hits, t_hits) of query a vs target b; both are concatenated 32-byte frame hashes."""
    assert len(a) % 32 == 0 and len(b) % 32 == 0
    aa = np.frombuffer(bytes(a), dtype=np.uint8)
    bb = np.frombuffer(bytes(b), dtype=np.uint8)
    q, t = C.c_int32(0), C.c_int32(0)
    rc = lib().hvd_cpu_match_two(aa.ctypes.data if aa.size else None, len(a) // 32,
                                 bb.ctypes.data if bb.size else None, len(b) // 32, max_dist, C.byref(q), C.byref(t))
    if rc != 0:
        raise RuntimeError(f"oracle match_two rc={rc}")
    return q.value, t.value


def match_videos(frames: np.ndarray, offsets: np.ndarray, max_dist: int = 31, cap: int = 1 << 20,
                 num_threads: int = 1) -> np.ndarray:
    """All video pairs a<b with >=1 frame hit; structured array VMATCH_DTYPE sorted by (a,b). num_threads > 1: the same
    per-pair statement on several host threads (rows dealt round-robin), identical output."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, 32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    V = offsets.size - 1
    out = np.zeros(max(cap, 1), dtype=VMATCH_DTYPE)
    cnt = C.c_int64(0)
    if num_threads > 1:
        rc = lib().hvd_cpu_vpdq_match_videos_mt(frames.ctypes.data, offsets.ctypes.data, V, max_dist, out.ctypes.data,
                                                cap, C.byref(cnt), num_threads)
    else:
        rc = lib().hvd_cpu_vpdq_match_videos(frames.ctypes.data, offsets.ctypes.data, V, max_dist, out.ctypes.data, cap,
                                             C.byref(cnt))
    if rc == -3:
        return match_videos(frames, offsets, max_dist, cap=int(cnt.value), num_threads=num_threads)
    if rc != 0:
        raise RuntimeError(f"oracle match_videos rc={rc}")
    return out[: cnt.value].copy()
