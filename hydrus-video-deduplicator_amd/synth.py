"""Deterministic synthetic inputs for tests and bench.py.

The reference's own fixtures (tests/testdb submodule: clips, known-good hashes) are
not available, so every workload is generated from a seed with numpy's PCG64.
"""

from __future__ import annotations

import numpy as np


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def frames_gray(n: int, seed: int = 2, h: int = 64, w: int = 64, const_fraction: float = 0.05) -> np.ndarray:
    """uint8[n,h,w]: smooth random fields (8 low-frequency cosines) + noise of varying
    amplitude, so that PDQ quality spans 0..100; a few exact constants (quality 0)."""
    rng = _rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32) / h, np.arange(w, dtype=np.float32) / w, indexing="ij")
    out = np.empty((n, h, w), dtype=np.uint8)
    chunk = 512
    for f0 in range(0, n, chunk):
        m = min(chunk, n - f0)
        img = np.full((m, h, w), 128.0, dtype=np.float32)
        amp_scale = rng.choice(np.array([0.05, 0.2, 1.0, 1.0], dtype=np.float32), size=m)
        for _ in range(8):
            fx = rng.uniform(0, 4, m).astype(np.float32)[:, None, None]
            fy = rng.uniform(0, 4, m).astype(np.float32)[:, None, None]
            ph = rng.uniform(0, 2 * np.pi, m).astype(np.float32)[:, None, None]
            amp = (rng.uniform(5, 40, m).astype(np.float32) * amp_scale)[:, None, None]
            img += amp * np.cos(2 * np.pi * (fx * xx[None] + fy * yy[None]) + ph)
        noise_amp = rng.choice(np.array([0.0, 1.0, 2.0, 4.0, 16.0], dtype=np.float32), size=m)[:, None, None]
        img += rng.uniform(-1, 1, (m, h, w)).astype(np.float32) * noise_amp
        frames = np.clip(img, 0, 255).astype(np.uint8)
        const = rng.random(m) < const_fraction
        vals = rng.integers(0, 256, m, dtype=np.uint8)
        frames[const] = vals[const][:, None, None]
        out[f0 : f0 + m] = frames
    return out


def frames_rgb(n: int, seed: int = 6, h: int = 512, w: int = 512) -> np.ndarray:
    """uint8[n,h,w,3] packed RGB24: three differently shifted copies of a gray field."""
    g = frames_gray(n, seed, h, w, const_fraction=0.0)
    rng = _rng(seed + 1000)
    out = np.empty((n, h, w, 3), dtype=np.uint8)
    out[..., 0] = g
    out[..., 1] = np.roll(g, 3, axis=2)
    out[..., 2] = 255 - np.roll(g, 5, axis=1)
    # a little per-channel noise so that R, G, B are not functions of each other
    out ^= rng.integers(0, 4, out.shape, dtype=np.uint8)
    return out


def flip_bits(rows: np.ndarray, k: np.ndarray, rng: np.random.Generator) -> np.ndarray:
    """Flip exactly k[r] distinct random bits of each 32-byte row."""
    rows = rows.copy()
    m = rows.shape[0]
    # argsort of random keys = random permutation of the 256 bit positions per row
    perm = np.argsort(rng.random((m, 256)), axis=1)
    sel = np.arange(256)[None, :] < k[:, None]
    flip = np.zeros((m, 256), dtype=np.uint8)
    np.put_along_axis(flip, perm, sel.astype(np.uint8), axis=1)
    rows ^= np.packbits(flip, axis=1, bitorder="little")
    return rows


def hash_db(n: int, seed: int = 3, plant_fraction: float = 0.001, max_flips: int = 40):
    """uint8[n,32] uniform random hashes with planted near-duplicates: plant_fraction of
    the rows become a copy of a random earlier row with k~U{0..max_flips} bit flips, so
    planted distances straddle the tolerance 31. Returns (db, planted) where planted is
    int64[m,3] = (src_row, dst_row, flips)."""
    rng = _rng(seed)
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    m = int(round(n * plant_fraction))
    if n < 2 or m == 0:
        return db, np.zeros((0, 3), dtype=np.int64)
    dst = np.sort(rng.choice(np.arange(1, n), size=min(m, n - 1), replace=False))
    src = (rng.random(dst.size) * dst).astype(np.int64)  # uniform in [0, dst)
    k = rng.integers(0, max_flips + 1, dst.size)
    # sources must not themselves be overwritten later: resolve in order
    for s, d, kk in zip(src, dst, k):
        db[d] = flip_bits(db[s : s + 1], np.array([kk]), rng)[0]
    return db, np.stack([src, dst, k], axis=1).astype(np.int64)


def video_hashes(num_videos: int, seed: int = 5, frames_per_video=64, copy_fraction: float = 0.02,
                 max_flips: int = 24):
    """Synthetic library of video hashes: (frames uint8[sum,32], offsets int64[V+1],
    planted int64[m,2]). frames_per_video: int or (lo, hi) for ragged lengths (0 allowed).
    copy_fraction of the videos are near-copies of an earlier video: every frame (or only
    the first half, for every other copy) is the source frame with up to max_flips flips."""
    rng = _rng(seed)
    if isinstance(frames_per_video, int):
        lens = np.full(num_videos, frames_per_video, dtype=np.int64)
    else:
        lo, hi = frames_per_video
        lens = rng.integers(lo, hi + 1, num_videos).astype(np.int64)
    offsets = np.zeros(num_videos + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    frames = rng.integers(0, 256, (int(offsets[-1]), 32), dtype=np.uint8)
    m = int(round(num_videos * copy_fraction))
    planted = []
    if num_videos >= 2 and m > 0:
        dst = np.sort(rng.choice(np.arange(1, num_videos), size=min(m, num_videos - 1), replace=False))
        for idx, d in enumerate(dst):
            s = int(rng.integers(0, d))
            ls, ld = int(lens[s]), int(lens[d])
            ncopy = min(ls, ld)
            if idx % 2 == 1:
                ncopy //= 2
            if ncopy == 0:
                continue
            k = rng.integers(0, max_flips + 1, ncopy)
            frames[offsets[d] : offsets[d] + ncopy] = flip_bits(frames[offsets[s] : offsets[s] + ncopy], k, rng)
            planted.append((s, int(d)))
    return frames, offsets, np.array(planted, dtype=np.int64).reshape(-1, 2)


def hash_db_clustered(n: int, n_clusters: int, cluster_size: int, seed: int = 8, max_flips: int = 8):
    """uint8[n,32] uniform random hashes in which n_clusters * cluster_size rows, scattered uniformly over the DB,
    form clusters of near-identical hashes (a random centre with k~U{0..max_flips} bit flips per member, so all
    members of a cluster are within 2*max_flips of each other): the regime of real frame hashes (static scenes,
    re-encodes), where many panels of the all-pairs kernel contain a hit. Returns (db, members int64[n_clusters,
    cluster_size]); the exact pair count is n_clusters * C(cluster_size, 2) when 2*max_flips <= tolerance."""
    rng = _rng(seed)
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    m = n_clusters * cluster_size
    if m > n:
        raise ValueError("clusters do not fit in the DB")
    pos = rng.choice(n, size=m, replace=False).astype(np.int64)
    centres = rng.integers(0, 256, (n_clusters, 32), dtype=np.uint8)
    rows = np.repeat(centres, cluster_size, axis=0)
    chunk = 1 << 16
    for c0 in range(0, m, chunk):
        sl = slice(c0, min(m, c0 + chunk))
        db[pos[sl]] = flip_bits(rows[sl], rng.integers(0, max_flips + 1, rows[sl].shape[0]), rng)
    return db, pos.reshape(n_clusters, cluster_size)
