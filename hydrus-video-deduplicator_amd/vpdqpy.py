"""Host-side mirror of the reference facade ``vpdqpy/vpdqpy.py`` (class Vpdq), bound to the
MI355X kernels instead of ``hvdaccelerators``. Same names, argument meaning and error
behaviour; video *decoding* (PyAV/FFmpeg, vpdqpy.py:58-101) is out of scope, so
``computeHash`` takes pre-decoded frames (or an iterable of frame byte strings) in the
format ``frame_extract_pyav`` yields: 512x512 packed rgb24 (vpdqpy.py:90-95)."""

from __future__ import annotations

import os
from collections.abc import Iterable

import numpy as np

from . import vpdq

# The dimensions of the image after downscaling for pdq (vpdqpy.py:23)
DOWNSCALE_DIMENSIONS = 512

VpdqHash = vpdq.VpdqHash


def hashed_frame_stride(average_rate) -> int:
    """Every how-many-th decoded frame the reference hashes (vpdqpy.py:72-77): ``round(average_rate)`` of the
    stream's average frame rate -- Python's round, so a Fraction of exactly k + 1/2 goes to the even neighbour,
    as it does there -- and 1 (every frame) when the rate is None or below 1 (small GIFs)."""
    if average_rate is None or average_rate < 1:
        return 1
    return round(average_rate)


def select_frames(frames, average_rate, bad_frame_errors: tuple = ()):
    """The frame-selection rule of ``frame_extract_pyav`` (vpdqpy.py:72-77,85-101) for a decoder-side caller:
    yield the frames whose DECODE index is a multiple of ``hashed_frame_stride(average_rate)``.

    `frames` is any iterable of decoded frames in decode order. An exception of a type listed in
    `bad_frame_errors` (the reference: ``av.error.InvalidDataError``) raised while fetching a frame skips that
    frame but still advances the index (vpdqpy.py:99-101), so the frames after it keep their phase. Which frames
    are hashed is part of the video hash: feed ``Vpdq.computeHash`` / ``VideoHasher.hash_frame`` from this."""
    stride = hashed_frame_stride(average_rate)
    it = iter(frames)
    frame_index = 0
    while True:
        try:
            frame = next(it)
            if frame_index % stride == 0:
                yield frame
            frame_index += 1
        except StopIteration:
            break
        except bad_frame_errors:
            frame_index += 1


def selected_frame_indices(n_decoded: int, average_rate) -> np.ndarray:
    """Decode indices `select_frames` keeps out of n_decoded good frames (array form: ``frames[idx]``)."""
    return np.arange(0, max(0, int(n_decoded)), hashed_frame_stride(average_rate), dtype=np.int64)


class Vpdq:
    @staticmethod
    def match_hash(query_features: VpdqHash, target_features: VpdqHash, distance_tolerance: float = 31.0):
        """Get the similarity of two videos by comparing their list of features (vpdqpy.py:49-56)."""
        return vpdq.matchHash(query_features, target_features, int(distance_tolerance))

    select_frames = staticmethod(select_frames)

    @staticmethod
    def computeHash(frames, num_threads: int = 0, width: int | None = None, height: int | None = None,
                    average_rate=None, all_decoded_frames: bool = False) -> VpdqHash:
        """Perceptually hash a video given its decoded frames (vpdqpy.py:103-119 minus decode).

        frames: uint8[n,h,w,3] / uint8[n,h,w] array, or an iterable of per-frame byte strings
        (then width/height default to DOWNSCALE_DIMENSIONS, as the reference passes). By default `frames` are
        the frames to hash (what ``frame_extract_pyav`` yields); with ``all_decoded_frames=True`` they are EVERY
        decoded frame and the reference's selection rule is applied first (``select_frames(frames, average_rate)``)."""
        if frames is None:
            raise ValueError
        if all_decoded_frames and not isinstance(frames, (bytes, bytearray, memoryview, str, os.PathLike)):
            frames = (frames[selected_frame_indices(frames.shape[0], average_rate)] if isinstance(frames, np.ndarray)
                      else select_frames(frames, average_rate))
        if isinstance(frames, (bytes, bytearray, memoryview, str, os.PathLike)):
            # the reference's caller passes the ENCODED video (dedup.py:76) and decodes it with PyAV; decoding is
            # out of scope here, and iterating a bytes object would silently hash garbage
            raise ValueError("encoded video input (bytes / path) is not supported: decode first and pass the frames "
                             "(uint8[n,h,w,3] array or an iterable of per-frame byte strings)")
        average_fps = 1  # timestamps are discarded (vpdqpy.py:110-112)
        if isinstance(frames, np.ndarray):
            if frames.ndim not in (3, 4):
                raise ValueError("frames must be uint8[n,h,w] or uint8[n,h,w,3]")
            h, w = frames.shape[1], frames.shape[2]
            hasher = vpdq.VideoHasher(average_fps, w, h, num_threads)
            flat = np.ascontiguousarray(frames, dtype=np.uint8).reshape(frames.shape[0], -1)
            for f in flat:
                hasher.hash_frame(f.data)
            return hasher.finish()
        if isinstance(frames, Iterable):
            w = DOWNSCALE_DIMENSIONS if width is None else width
            h = DOWNSCALE_DIMENSIONS if height is None else height
            hasher = vpdq.VideoHasher(average_fps, w, h, num_threads)
            for frame in frames:
                hasher.hash_frame(frame)
            return hasher.finish()
        raise ValueError("Failed to hash: invalid frames object type.")

    @staticmethod
    def is_similar(vpdq_features1: VpdqHash, vpdq_features2: VpdqHash, threshold: float = 75.0) -> tuple[bool, float]:
        """Threshold is minimum similarity to be considered similar (vpdqpy.py:121-131)."""
        similarity = Vpdq.match_hash(query_features=vpdq_features1, target_features=vpdq_features2)
        return similarity >= threshold, similarity
