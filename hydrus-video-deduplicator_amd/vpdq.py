"""Drop-in for the native module ``hvdaccelerators.vpdq`` (PyPI hvdaccelerators==0.4.0),
i.e. exactly the five names the reference imports from it:

    vpdq.VideoHasher      vpdqpy/vpdqpy.py:113      VideoHasher(fps, w, h, num_threads)
      .hash_frame(bytes)  vpdqpy/vpdqpy.py:118
      .finish()->VpdqHash vpdqpy/vpdqpy.py:119
    vpdq.VpdqHash         vpdqpy/vpdqpy.py:25       .bytes (dedup.py:77), bytesPerPdqHash
                                                     (dedup.py:83), str() (hashing.py:30),
                                                     from_string (hashing.py:40), len/==/!=
                                                     (tests/unit_tests/test_vpdqpy.py:95,116)
    vpdq.matchHash        vpdqpy/vpdqpy.py:56       matchHash(q, t, tol) -> float in [0,100]
    vpdq.matchHashBytes   db/vptree.py:31           matchHashBytes(a, b, tol)

All arithmetic runs in HIP kernels on an MI355X through the C-ABI of
include/hvd_mi355x.h; there is no CPU fallback.

Semantics that the (absent) wheel does not let us pin are explicit, tested policies
(SURVEY.md 3.5), each selectable by an environment variable or at run time:

* ``MATCH_COMPARATOR`` (``HVD_MATCH_COMPARATOR`` = ``le`` | ``lt``): a frame pair is a hit when
  ``hamming <= tolerance`` (default) or ``hamming < tolerance``. Distances are integers, so ``lt``
  is ``le`` at ``tolerance - 1``; the kernels always take the inclusive bound (``frame_max_dist``).
  UNVERIFIED DEFAULT: upstream's matchTwoHashBrute may be strict; tests/golden/import_reference.py
  settles it where the wheel is installable.
* ``MATCH_POLICY`` (``HVD_MATCH_POLICY``): reduction of the two vPDQ percentages (query-matched %,
  target-matched %) to one number; default "min", the only symmetric choice, which
  dedup.py:502's ``// 2`` assumes.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

BYTES_PER_PDQ_HASH = 32
QUALITY_TOLERANCE = 31  # frames with quality >= 31 are kept (db/DedupeDB.py:550-553)

# "min" | "max" | "query" | "target"; see module docstring.
MATCH_POLICY = os.environ.get("HVD_MATCH_POLICY", "min")
_POLICIES = ("min", "max", "query", "target")

# "le" | "lt"; see module docstring.
MATCH_COMPARATOR = os.environ.get("HVD_MATCH_COMPARATOR", "le")
_COMPARATORS = ("le", "lt")


def policy_labels() -> dict:
    """The semantic switches the absent `hvdaccelerators` wheel leaves open, as labels for every artefact that reports
    results (bench.py's JSON line, hvd_runtime_info): `(unverified)` marks a DEFAULT that nothing reference-held confirms;
    a value set explicitly through the environment is the caller's decision and carries no mark."""
    return {"comparator": MATCH_COMPARATOR + ("" if "HVD_MATCH_COMPARATOR" in os.environ else " (unverified)"),
            "reduction": MATCH_POLICY + ("" if "HVD_MATCH_POLICY" in os.environ else " (unverified)"),
            "dct": "strict" if _lib_dct_mode() == 0 else "fma",
            "hash_text": "hex, 64 characters per frame (unverified)"}


def _lib_dct_mode() -> int:
    try:
        return int(_lib.load().hvd_get_pdq_dct_mode())
    except Exception:  # noqa: BLE001 - labels must be printable without the library
        return 0 if os.environ.get("HVD_PDQ_DCT_MODE", "strict") != "fma" else 1


_warned_unverified = False


def warn_unverified_policies() -> None:
    """One RuntimeWarning per process, the first time a matchHash* entry runs with the comparator left at its unverified
    default: upstream's matchTwoHashBrute may compare with a strict `<`."""
    global _warned_unverified
    if _warned_unverified or "HVD_MATCH_COMPARATOR" in os.environ:
        return
    _warned_unverified = True
    import warnings

    warnings.warn("hvd_amd: frame comparator left at its default 'le' (a frame pair at Hamming distance == tolerance is a hit); "
                  "hvdaccelerators 0.4.0 could not be consulted and upstream vPDQ may use a strict '<'. Set "
                  "HVD_MATCH_COMPARATOR=le|lt to state the choice (INTEGRATION.md section 4).", RuntimeWarning, stacklevel=3)


def frame_max_dist(distance_tolerance, comparator: str | None = None) -> int:
    """Inclusive Hamming bound the kernels use for a reference-style tolerance under the comparator
    policy: tolerance for "le", tolerance - 1 for "lt" (-1 = nothing matches)."""
    comparator = MATCH_COMPARATOR if comparator is None else comparator
    if comparator not in _COMPARATORS:
        raise ValueError(f"unknown comparator {comparator!r}; expected one of {_COMPARATORS}")
    tol = int(distance_tolerance)
    return tol if comparator == "le" else tol - 1


def set_dct_mode(mode: str) -> None:
    """"strict" (default): separately rounded multiply and add, the numerics of upstream's x86-64
    builds; "fma": fused multiply-add on the matrix cores, the numerics of upstream's arm64 builds
    (opt-in; ~2 % of hash bits differ between the two). See include/hvd_mi355x.h."""
    modes = {"strict": 0, "fma": 1}
    if mode not in modes:
        raise ValueError(f"unknown DCT mode {mode!r}; expected one of {sorted(modes)}")
    _lib.check(_lib.load().hvd_set_pdq_dct_mode(modes[mode]))


def get_dct_mode() -> str:
    return ("strict", "fma")[_lib.load().hvd_get_pdq_dct_mode()]


def percent_from_hits(q_hits: int, t_hits: int, nq: int, nt: int, policy: str | None = None) -> float:
    """vPDQ match percentage from the kernel's two counters (either side empty -> 0.0,
    db/DedupeDB.py:555-557)."""
    policy = MATCH_POLICY if policy is None else policy
    if policy not in _POLICIES:
        raise ValueError(f"unknown match policy {policy!r}; expected one of {_POLICIES}")
    if nq <= 0 or nt <= 0:
        return 0.0
    qp = (q_hits * 100.0) / nq
    tp = (t_hits * 100.0) / nt
    if policy == "min":
        return min(qp, tp)
    if policy == "max":
        return max(qp, tp)
    return qp if policy == "query" else tp


class VpdqHash:
    """Immutable value: N concatenated 32-byte PDQ frame hashes (N >= 0)."""

    bytesPerPdqHash = BYTES_PER_PDQ_HASH
    __slots__ = ("_b",)

    def __init__(self, data: bytes = b""):
        data = bytes(data)
        if len(data) % BYTES_PER_PDQ_HASH != 0:
            raise ValueError(f"VpdqHash needs a multiple of {BYTES_PER_PDQ_HASH} bytes, got {len(data)}")
        self._b = data

    @property
    def bytes(self) -> bytes:
        return self._b

    def __len__(self) -> int:
        return len(self._b) // BYTES_PER_PDQ_HASH

    def __eq__(self, other) -> bool:
        return isinstance(other, VpdqHash) and self._b == other._b

    def __ne__(self, other) -> bool:
        return not self.__eq__(other)

    def __hash__(self) -> int:
        return hash(self._b)

    def __str__(self) -> str:
        # contiguous lowercase hex of .bytes, 64 chars per frame, single line
        # (db/DedupeDB.py:547-559; tests read it back with readline(), test_benchmark_vpdqpy.py:58-59)
        return self._b.hex()

    def __repr__(self) -> str:
        return f"VpdqHash(frames={len(self)})"

    @staticmethod
    def from_string(s: str) -> "VpdqHash":
        s = s.strip()
        if len(s) % (2 * BYTES_PER_PDQ_HASH) != 0:
            raise ValueError("VpdqHash string must be 64 hex characters per frame")
        try:
            return VpdqHash(bytes.fromhex(s))
        except ValueError as exc:
            raise ValueError(f"invalid VpdqHash string: {exc}") from exc

    def frames(self) -> np.ndarray:
        """uint8[N,32] view of the frame hashes."""
        return np.frombuffer(self._b, dtype=np.uint8).reshape(-1, BYTES_PER_PDQ_HASH)


class VideoHasher:
    """Per-video frame hasher: the Python face of the native streaming hasher (hvd_hasher_*,
    csrc/hvd_stream.cpp). Frames are copied once into a ring of pinned batch slots; each batch is
    uploaded, hashed and downloaded on its own HIP stream, so PCIe transfer overlaps the PDQ
    kernels. ``hash_frame`` blocks only when every slot is still in flight, which bounds the
    staging memory like the reference's blocking frame queue (vpdqpy/vpdqpy.py:115-117). The
    reference's hasher runs a CPU thread pool instead; ``num_threads`` is accepted for signature
    compatibility; what it controls here is how many host threads share the copy of one frame into the ring
    (0 = library default). One hasher per decoder thread."""

    def __init__(self, average_fps: int, width: int, height: int, num_threads: int = 0,
                 batch_bytes: int = 32 << 20):
        if width < 64 or height < 64:
            raise ValueError("frames must be at least 64x64")
        self.average_fps = average_fps
        self.width = int(width)
        self.height = int(height)
        self.num_threads = num_threads
        self._frame_bytes_rgb = self.width * self.height * 3
        self._frame_bytes_gray = self.width * self.height
        self._batch_bytes = int(batch_bytes)
        self._channels = 0
        self._handle = None
        self._finished = False
        self._run = None        # acquire_frame(): view of the run of frames acquired from the native hasher ...
        self._run_pos = 0       # ... and how many of them commit_frame() has counted
        self._batch_run = None  # acquire_frames(): the run handed out
        self._lib = _lib.ensure()  # fail at construction, not at the first frame, if no GPU is usable

    def _open(self, channels: int) -> None:
        frame_bytes = self._frame_bytes_rgb if channels == 3 else self._frame_bytes_gray
        batch = max(1, min(4096, self._batch_bytes // frame_bytes))
        h = C.c_void_p()
        _lib.check(self._lib.hvd_hasher_create(self.width, self.height, channels, batch, C.byref(h)))
        self._handle = h
        self._channels = channels
        # num_threads: the reference hasher's worker threads (vpdqpy/vpdqpy.py:113; 0 = library default, negative =
        # "all but n cores", entrypoint.py:79-82) -> the threads that share the host-side copy of a frame into the ring
        nt = int(self.num_threads) if isinstance(self.num_threads, int) else 0
        if nt < 0:
            nt = max(1, (os.cpu_count() or 1) + nt)
        _lib.check(self._lib.hvd_hasher_set_threads(h, nt))

    def _check_feed(self, channels: int) -> None:
        if self._finished:
            raise RuntimeError("frame fed after finish()")
        if channels not in (1, 3):
            raise ValueError("channels must be 1 (gray) or 3 (rgb24)")
        if self._handle is None:
            self._open(channels)
        elif channels != self._channels:
            raise ValueError("all frames of one video must have the same pixel format")

    def _acquire_run(self, want: int) -> np.ndarray:
        """hvd_hasher_acquire_n: one view [got, h, w(, 3)] over the next `got` <= want frames of the pinned slot."""
        p, got = C.c_void_p(), C.c_int64(0)
        _lib.check(self._lib.hvd_hasher_acquire_n(self._handle, int(want), C.byref(p), C.byref(got)))
        ch = self._channels
        fb = self._frame_bytes_rgb if ch == 3 else self._frame_bytes_gray
        buf = (C.c_uint8 * (fb * got.value)).from_address(p.value)
        shape = (got.value, self.height, self.width, 3) if ch == 3 else (got.value, self.height, self.width)
        return np.frombuffer(buf, dtype=np.uint8).reshape(shape)

    def _flush_run(self) -> None:
        """Hand the frames committed one by one (commit_frame) to the native hasher: one hvd_hasher_commit_n per run."""
        if self._run is not None:
            n, self._run = self._run_pos, None
            self._run_pos = 0
            _lib.check(self._lib.hvd_hasher_commit_n(self._handle, n))

    def acquire_frames(self, k: int, channels: int = 3) -> np.ndarray:
        """Zero-copy feed for a RUN of frames (hvd_hasher_acquire_n): a writable uint8 view ``[got, height, width(, 3)]``
        of the pinned slot memory for the next frames, ``1 <= got <= k`` (what is left of the current batch slot, so a
        caller loops until its frames are placed); ``commit_frames(n)`` makes the first n count. One FFI round trip per
        run: a decoder of small frames (64x64) fills hundreds of frames per call. Blocks like ``hash_frame`` when every
        batch slot is in flight (vpdqpy/vpdqpy.py:115-117)."""
        if k < 1:
            raise ValueError("k must be at least 1")
        self._check_feed(channels)
        self._flush_run()
        self._batch_run = self._acquire_run(k)
        return self._batch_run

    def commit_frames(self, n: int | None = None) -> None:
        """The first n frames of the run ``acquire_frames`` handed out are complete (default: all of them)."""
        if self._batch_run is None:
            raise RuntimeError("commit_frames() without acquire_frames()")
        n = self._batch_run.shape[0] if n is None else int(n)
        self._batch_run = None
        _lib.check(self._lib.hvd_hasher_commit_n(self._handle, n))

    def acquire_frame(self, channels: int = 3) -> np.ndarray:
        """Zero-copy feed (hvd_hasher_acquire_n underneath): a writable uint8 view of the pinned slot memory for the NEXT
        frame -- shape (height, width, 3) or (height, width) -- so that a decoder can reformat straight into
        it (e.g. ``frame.to_ndarray(...)`` with ``out=``, or ``np.copyto``); ``commit_frame()`` makes it count.
        Blocks like ``hash_frame`` when every batch slot is in flight. The frames of one batch slot are handed out from ONE
        native call and committed with one (round 5: two ctypes calls and a buffer object per frame made this feed slower
        than ``hash_frame(bytes)`` for 64x64 frames)."""
        if self._run is None or self._run_pos == self._run.shape[0]:
            self._check_feed(channels)
            self._flush_run()
            self._run = self._acquire_run(1 << 30)  # the rest of the current batch slot
        elif channels != self._channels:
            raise ValueError("all frames of one video must have the same pixel format")
        return self._run[self._run_pos]

    def commit_frame(self) -> None:
        if self._run is None:
            raise RuntimeError("commit_frame() without acquire_frame()")
        self._run_pos += 1
        if self._run_pos == self._run.shape[0]:
            self._flush_run()  # the batch slot is full: submitted (H2D + kernels) by the native side

    def hash_frame(self, frame) -> None:
        """frame: packed RGB24 bytes (width*height*3, what bytes(frame.planes[0]) yields at
        vpdqpy.py:118) or gray bytes (width*height). The buffer is copied before returning."""
        if self._finished:
            raise RuntimeError("hash_frame() after finish()")
        if isinstance(frame, bytes):
            n, ptr, keep = len(frame), frame, frame
        else:
            keep = None
            if isinstance(frame, np.ndarray) and frame.flags.c_contiguous and frame.flags.writeable:
                # (0.3 us; `.ctypes.data` builds a helper object per call: 1.1 us of the 1 us a 64x64 frame costs otherwise)
                keep, n = frame, frame.nbytes
                ptr = C.addressof(C.c_uint8.from_buffer(frame))
            if keep is None:
                keep = np.frombuffer(frame, dtype=np.uint8)  # zero-copy view of any buffer object
                n, ptr = keep.size, keep.ctypes.data
        if n == self._frame_bytes_rgb:
            ch = 3
        elif n == self._frame_bytes_gray:
            ch = 1
        else:
            raise ValueError(f"frame has {n} bytes; expected {self._frame_bytes_rgb} (rgb24) or "
                             f"{self._frame_bytes_gray} (gray)")
        if self._handle is None:
            self._open(ch)
        elif ch != self._channels:
            raise ValueError("all frames of one video must have the same pixel format")
        if self._run is not None:
            self._flush_run()
        _lib.check(self._lib.hvd_hasher_push(self._handle, ptr))
        del keep

    def finish(self) -> VpdqHash:
        self._finished = True
        if self._handle is None:
            return VpdqHash(b"")
        try:
            self._flush_run()
            self._batch_run = None
            pending = C.c_int64(0)
            _lib.check(self._lib.hvd_hasher_pending(self._handle, C.byref(pending)))
            n = pending.value
            hashes = np.zeros((max(n, 1), BYTES_PER_PDQ_HASH), dtype=np.uint8)
            quality = np.zeros(max(n, 1), dtype=np.int32)
            got = C.c_int64(0)
            _lib.check(self._lib.hvd_hasher_finish(self._handle, hashes.ctypes.data, quality.ctypes.data, n,
                                                   C.byref(got)))
            assert got.value == n
            hashes, quality = hashes[:n], quality[:n]
            return VpdqHash(hashes[quality >= QUALITY_TOLERANCE].tobytes())
        finally:
            self.close()

    def close(self) -> None:
        if self._handle is not None:
            self._lib.hvd_hasher_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hash_frames(frames: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Batch entry point: uint8[n,h,w] (gray) or uint8[n,h,w,3] (rgb24) ->
    (hashes uint8[n,32], quality int32[n]). No quality filtering."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    lib = _lib.ensure()
    if frames.ndim == 3:
        fn = lib.hvd_pdq_hash_frames_gray_u8
    elif frames.ndim == 4 and frames.shape[3] == 3:
        fn = lib.hvd_pdq_hash_frames_rgb24_u8
    else:
        raise ValueError("frames must be uint8[n,h,w] or uint8[n,h,w,3]")
    n, h, w = frames.shape[:3]
    hashes = np.zeros((n, BYTES_PER_PDQ_HASH), dtype=np.uint8)
    quality = np.zeros(n, dtype=np.int32)
    _lib.check(fn(frames.ctypes.data, n, h, w, hashes.ctypes.data, quality.ctypes.data))
    return hashes, quality


def match_counts(a: bytes, b: bytes, distance_tolerance: int = 31) -> tuple[int, int]:
    """(q_hits, t_hits) for query a / target b, both concatenated 32-byte frame hashes."""
    a = bytes(a)
    b = bytes(b)
    if len(a) % BYTES_PER_PDQ_HASH or len(b) % BYTES_PER_PDQ_HASH:
        raise ValueError("hash byte strings must be multiples of 32 bytes")
    na, nb = len(a) // BYTES_PER_PDQ_HASH, len(b) // BYTES_PER_PDQ_HASH
    lib = _lib.ensure()
    q, t = C.c_int32(0), C.c_int32(0)
    _lib.check(lib.hvd_match_two(a if na else None, na, b if nb else None, nb, int(distance_tolerance),
                                 C.byref(q), C.byref(t)))
    return q.value, t.value


def matchHashBytes(a: bytes, b: bytes, distance_tolerance: int) -> float:
    """db/vptree.py:31 call shape: similarity in [0,100] from two raw BLOBs."""
    if not _warned_unverified:
        warn_unverified_policies()
    max_dist = frame_max_dist(distance_tolerance)
    if max_dist < 0:
        if len(a) % BYTES_PER_PDQ_HASH or len(b) % BYTES_PER_PDQ_HASH:
            raise ValueError("hash byte strings must be multiples of 32 bytes")
        return 0.0
    q, t = match_counts(a, b, max_dist)
    return percent_from_hits(q, t, len(a) // BYTES_PER_PDQ_HASH, len(b) // BYTES_PER_PDQ_HASH)


def matchHash(query: VpdqHash, target: VpdqHash, distance_tolerance: int) -> float:
    """vpdqpy/vpdqpy.py:56 call shape."""
    return matchHashBytes(query.bytes, target.bytes, distance_tolerance)
