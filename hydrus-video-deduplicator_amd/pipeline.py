"""Batch pipeline over a whole library: hash every video's frames in one pass over HBM, then
search all video pairs (BASELINE config 5 shape). The reference does this one video at a time
(`dedup.py:346-352`) and one tree probe per video (`dedup.py:468-475`); a GPU wants the batch.
"""

from __future__ import annotations

import numpy as np

from . import search, vpdq


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
