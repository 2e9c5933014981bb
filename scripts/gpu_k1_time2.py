"""Dev probe: product-default 64x64 hash launch (grid / operand rule of launch_pdq_hash64) at several batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
fr = synth.frames_gray(10000, seed=2)
for nf in [int(x) for x in os.environ.get("NS", "1000,4000,8192,10000,20000,65536,400000").split(",")]:
    d_f = L.DeviceBuffer(nf * 4096)
    for r0 in range(0, nf, 10000):
        m = min(10000, nf - r0)
        L.check(lib.hvd_memcpy_h2d(C.c_void_p(d_f.ptr + r0 * 4096), fr.ctypes.data, m * 4096))
    d_h, d_q = L.DeviceBuffer(32 * nf), L.DeviceBuffer(4 * nf)
    for grid in [int(g) for g in os.environ.get("GRIDS", "0").split(",")]:
        L.check(lib.hvd_debug_set(b"pdq_hash_grid", grid))
        ks = []
        for r in range(int(os.environ.get("REPS", 40))):
            L.check(lib.hvd_timer_start())
            L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, nf, 64, 64, 1, None, d_h.ptr, d_q.ptr))
            ms = C.c_float(0)
            L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r >= int(os.environ.get("REPS", 40)) // 2:
                ks.append(ms.value)
        print(f"n={nf:7d} grid {grid:5d}: {np.mean(ks) * 1e3:9.2f} us  {nf / np.mean(ks) * 1e3:.4g} frames/s", flush=True)
    d_f.free(); d_h.free(); d_q.free()
