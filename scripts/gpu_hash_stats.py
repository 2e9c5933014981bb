"""Distance statistics of the synthetic PDQ frame hashes (dev probe): how often do unrelated frames pass a
128-bit prefilter, for different choices of the 128 bits?"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, pipeline

lib = L.init(0)
V, F = 400, 64
n = V * F
d_frames = L.DeviceBuffer(n * 4096)
L.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, V, F, 5, None))
d_h, d_q = pipeline.hash_frames_on_device(d_frames.ptr, n, 64, 64, 1)
h = d_h.to_array(np.uint8, 32 * n).reshape(-1, 32)
q = d_q.to_array(np.int32, n)
h = h[q >= 31]
print("kept", len(h), "quality hist", np.histogram(q, bins=[0, 1, 31, 50, 80, 101])[0])
bits = np.unpackbits(h, axis=1, bitorder="little")  # [n,256], bit k = i*16+j
print("bit balance (mean of set bits per position) min/max:", bits.mean(0).min(), bits.mean(0).max())
w = h.view(np.uint64)  # [n,4]
def popc(x): return np.bitwise_count(x).astype(np.int32)
i, j = np.divmod(np.arange(256), 16)
subsets = {
    "first128 (rows 0-7)": np.arange(256) < 128,
    "rows 0-3,8-11": ((i // 4) % 2) == 0,
    "even rows": (i % 2) == 0,
    "checkerboard": ((i + j) % 2) == 0,
    "even bits": (np.arange(256) % 2) == 0,
    "last128 (rows 8-15)": np.arange(256) >= 128,
    "cols 8-15": j >= 8,
    "high quadrant-ish (i+j>=15)": (i + j) >= 15,
}
masks = {}
for k, m in subsets.items():
    assert m.sum() in (128, 136), (k, m.sum())
    masks[k] = np.packbits(m.astype(np.uint8), bitorder="little").view(np.uint64)
N = len(h)
tot = 0
full_hits = 0
pf = {k: 0 for k in subsets}
hist = np.zeros(257, dtype=np.int64)
for r0 in range(0, N, 512):
    a = w[r0:r0 + 512][:, None, :]
    x = a ^ w[None, :, :]
    d = popc(x).sum(2)
    # different videos only is approximated by excluding |i-j| < 64
    idx = np.arange(r0, min(N, r0 + 512))[:, None]
    valid = np.abs(idx - np.arange(N)[None, :]) >= 64
    tot += valid.sum()
    full_hits += ((d <= 31) & valid).sum()
    hist += np.bincount(d[valid], minlength=257)
    for k, m in masks.items():
        dp = popc(x & m[None, None, :]).sum(2)
        pf[k] += ((dp <= 31) & valid).sum()
print("pairs", tot, "full<=31:", full_hits, f"({full_hits / tot:.3e})")
for k in subsets:
    print(f"  prefilter {k:32s} pass rate {pf[k] / tot:.3e}  -> P(panel-wave of 8192 pairs has one) {1 - np.exp(-8192 * pf[k] / tot):.3f}")
c = np.cumsum(hist) / tot
print("cumulative full-distance distribution: ", {d: f"{c[d]:.2e}" for d in (31, 40, 50, 62, 70, 80, 96, 110, 128)})
