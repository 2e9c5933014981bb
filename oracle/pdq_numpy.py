"""Second, independent CPU restatement of PDQ frame hashing in numpy float32.

TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (see oracle/hvd_oracle.c header): the real
arithmetic lives in the absent wheel ``hvdaccelerators==0.4.0`` (reference
pyproject.toml:36); this file restates the published ThreatExchange PDQ algorithm a
second time, structured differently from the C oracle (vectorised across the
independent axis, sort-based median instead of Torben selection), so that
bit-for-bit agreement of the two is evidence that neither has an accidental bug.

Call-site anchors in the reference: vpdqpy/vpdqpy.py:113-119 (hash_frame on rgb24),
db/DedupeDB.py:535-559 (32-byte little-endian layout, quality >= 31 kept).
"""

from __future__ import annotations

import math

import numpy as np

F = np.float32


def dct_matrix() -> np.ndarray:
    """16x64 float32; float scale * double cos rounded once (pdqhashing.cpp)."""
    scale = float(F(math.sqrt(2.0 / 64.0)))
    d = np.empty((16, 64), dtype=np.float32)
    for i in range(16):
        for j in range(64):
            d[i, j] = F(scale * math.cos((math.pi / 2 / 64.0) * (i + 1) * (2 * j + 1)))
    return d


_D = dct_matrix()


def luma_rgb(rgb: np.ndarray) -> np.ndarray:
    """rgb uint8[..., 3] -> float32[...]: ((0.299f*R + 0.587f*G) + 0.114f*B), each op rounded."""
    r = rgb[..., 0].astype(np.float32)
    g = rgb[..., 1].astype(np.float32)
    b = rgb[..., 2].astype(np.float32)
    y = F(0.299) * r
    y = y + F(0.587) * g
    y = y + F(0.114) * b
    return y.astype(np.float32)


def luma_gray(gray: np.ndarray) -> np.ndarray:
    v = gray.astype(np.float32)
    y = F(0.299) * v
    y = y + F(0.587) * v
    y = y + F(0.114) * v
    return y.astype(np.float32)


def _box_axis0(a: np.ndarray, w: int) -> np.ndarray:
    """Sequential running-sum box filter along axis 0 (box1DFloat), vectorised over axis 1."""
    n = a.shape[0]
    out = np.empty_like(a)
    half = (w + 2) // 2
    p1, p2, p3, p4 = half - 1, w - half + 1, n - w, half - 1
    s = np.zeros(a.shape[1], dtype=np.float32)
    li = ri = oi = 0
    cur = 0
    for _ in range(p1):
        s = s + a[ri]
        cur += 1
        ri += 1
    for _ in range(p2):
        s = s + a[ri]
        cur += 1
        out[oi] = s / F(cur)
        ri += 1
        oi += 1
    for _ in range(p3):
        s = s + a[ri]
        s = s - a[li]
        out[oi] = s / F(cur)
        li += 1
        ri += 1
        oi += 1
    for _ in range(p4):
        s = s - a[li]
        cur -= 1
        out[oi] = s / F(cur)
        li += 1
        oi += 1
    return out


def jarosz_decimate(luma: np.ndarray) -> np.ndarray:
    h, w = luma.shape
    win_rows = (w + 127) // 128  # window along a row, from the column count
    win_cols = (h + 127) // 128
    a = luma.astype(np.float32)
    for _ in range(2):
        a = _box_axis0(a.T.copy(), win_rows).T.copy()  # along rows
        a = _box_axis0(a, win_cols)  # along columns
    ii = [int(((i + 0.5) * h) / 64) for i in range(64)]
    jj = [int(((j + 0.5) * w) / 64) for j in range(64)]
    return a[np.ix_(ii, jj)].astype(np.float32)


def quality(a64: np.ndarray) -> int:
    dv = ((a64[:-1, :] - a64[1:, :]) * F(100.0)) / F(255.0)
    dh = ((a64[:, :-1] - a64[:, 1:]) * F(100.0)) / F(255.0)
    g = int(np.abs(np.trunc(dv).astype(np.int64)).sum() + np.abs(np.trunc(dh).astype(np.int64)).sum())
    return min(g // 90, 100)


def _fma32(a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Elementwise float32 fused multiply-add with ONE rounding, without calling libm: a*b is
    exact in float64 (24+24 significant bits); the exact sum of that product and c is obtained with
    an error-free TwoSum in float64 and rounded to float32 once, breaking double-rounding ties with
    the sign of the residual."""
    p = a.astype(np.float64) * b.astype(np.float64)
    c64 = c.astype(np.float64)
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)            # s + err == p + c exactly
    r = s.astype(np.float32)
    # s may sit exactly half-way between two floats while the true sum does not: nudge by the residual
    lo = np.nextafter(r, np.float32(-np.inf))
    hi = np.nextafter(r, np.float32(np.inf))
    mid_lo = (lo.astype(np.float64) + r.astype(np.float64)) / 2
    mid_hi = (hi.astype(np.float64) + r.astype(np.float64)) / 2
    r = np.where((s == mid_lo) & (err < 0), lo, r)   # rounded up from an exact tie but the true value is below it
    r = np.where((s == mid_hi) & (err > 0), hi, r)
    return r.astype(np.float32)


def dct16(a64: np.ndarray, fma: bool = False) -> np.ndarray:
    if fma:
        t = np.zeros((16, 64), dtype=np.float32)
        for k in range(64):
            t = _fma32(np.broadcast_to(_D[:, k : k + 1], (16, 64)), np.broadcast_to(a64[k : k + 1, :], (16, 64)), t)
        b = np.zeros((16, 16), dtype=np.float32)
        for k in range(64):
            b = _fma32(np.broadcast_to(t[:, k : k + 1], (16, 16)), np.broadcast_to(_D[:, k][None, :], (16, 16)), b)
        return b
    t = np.zeros((16, 64), dtype=np.float32)
    for k in range(64):
        t = t + _D[:, k : k + 1] * a64[k : k + 1, :]
    b = np.zeros((16, 16), dtype=np.float32)
    for k in range(64):
        b = b + t[:, k : k + 1] * _D[:, k][None, :]
    return b


def hash_from_luma(luma: np.ndarray, fma: bool = False) -> tuple[bytes, int, np.ndarray]:
    """-> (32-byte hash, quality, 16x16 DCT coefficients)."""
    h, w = luma.shape
    a64 = luma.astype(np.float32) if (h, w) == (64, 64) else jarosz_decimate(luma)
    q = quality(a64)
    b = dct16(a64, fma)
    med = np.sort(b.ravel())[127]  # Torben on 256 values returns the 128th smallest
    bits = (b.ravel() > med).astype(np.uint8)
    return np.packbits(bits, bitorder="little").tobytes(), q, b


def hash_gray64_batch(frames: np.ndarray, fma: bool = False):
    """uint8[n,64,64] -> (hashes u8[n,32], quality i32[n], coeffs f32[n,256]); the same arithmetic as
    hash_gray, vectorised over the frame axis so that 10^4 frames take seconds."""
    a = luma_gray(frames)                                                # [n,64,64]
    dv = ((a[:, :-1, :] - a[:, 1:, :]) * F(100.0)) / F(255.0)
    dh = ((a[:, :, :-1] - a[:, :, 1:]) * F(100.0)) / F(255.0)
    g = np.abs(np.trunc(dv).astype(np.int64)).sum((1, 2)) + np.abs(np.trunc(dh).astype(np.int64)).sum((1, 2))
    q = np.minimum(g // 90, 100).astype(np.int32)
    n = len(frames)
    t = np.zeros((n, 16, 64), dtype=np.float32)
    for k in range(64):
        dk, ak = _D[None, :, k : k + 1], a[:, k : k + 1, :]
        t = _fma32(np.broadcast_to(dk, t.shape), np.broadcast_to(ak, t.shape), t) if fma else t + dk * ak
    b = np.zeros((n, 16, 16), dtype=np.float32)
    for k in range(64):
        tk, dk = t[:, :, k : k + 1], _D[None, None, :, k]
        b = _fma32(np.broadcast_to(tk, b.shape), np.broadcast_to(dk, b.shape), b) if fma else b + tk * dk
    b = b.reshape(n, 256)
    med = np.sort(b, axis=1)[:, 127:128]
    bits = (b > med).astype(np.uint8)
    return np.packbits(bits, axis=1, bitorder="little"), q, b


def hash_gray(frame: np.ndarray, fma: bool = False):
    return hash_from_luma(luma_gray(frame), fma)


def hash_rgb(frame: np.ndarray, fma: bool = False):
    return hash_from_luma(luma_rgb(frame), fma)


def hamming(a: bytes, b: bytes) -> int:
    return int(np.unpackbits(np.frombuffer(a, np.uint8) ^ np.frombuffer(b, np.uint8)).sum())
