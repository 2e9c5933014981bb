"""Developer check: one-wave-per-frame 512x512 down-sampler (k_down512w) vs the workgroup form: parity + timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
from oracle import oracle as O
lib = L.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = synth.frames_rgb(16, seed=6)
ho, qo = O.hash_frames(base, num_threads=16)
gray = synth.frames_gray(8, seed=9, h=512, w=512)
hgo, qgo = O.hash_frames(gray, num_threads=8)
fr = np.concatenate([base] * (n // 16))
sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(n, 512, 512, 3, C.byref(sb)))
d_f = L.DeviceBuffer.from_array(fr); d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
grids = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2304]
for wave, grid in [(0, 0)] + [(2, g) for g in grids]:
    L.check(lib.hvd_debug_set(b"pdq_down512_wave", wave))
    if wave: L.check(lib.hvd_debug_set(b"pdq_down512_wave_grid", grid))
    hg, qg = hvd_amd.vpdq.hash_frames(gray)
    pg = np.array_equal(hg, hgo) and np.array_equal(qg, qgo)
    L.check(lib.hvd_dev_memset(d_s.ptr, 0xFF, sb.value)); L.check(lib.hvd_dev_memset(d_h.ptr, 0, 32 * n)); L.check(lib.hvd_dev_memset(d_q.ptr, 0xFF, 4 * n))
    best = 1e9
    for r in range(4):
        L.check(lib.hvd_timer_start())
        L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 512, 512, 3, d_s.ptr, d_h.ptr, d_q.ptr))
        ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
        if r: best = min(best, ms.value)
    h = d_h.to_array(np.uint8, 32 * n).reshape(-1, 32); q = d_q.to_array(np.int32, n)
    ok = np.array_equal(h, np.concatenate([ho] * (n // 16))) and np.array_equal(q, np.concatenate([qo] * (n // 16)))
    bad = int((h != np.concatenate([ho] * (n // 16))).any(1).sum())
    print(f"wave={wave} grid={grid} n={n}: gray parity {pg} rgb parity {ok} (bad frames {bad})  {best:.3f} ms  {n / best:.1f} kframes/s  ({n * 786468 / best / 1e6:.1f} GB/s algorithmic)", flush=True)
