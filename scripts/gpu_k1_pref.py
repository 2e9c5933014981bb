"""Dev probe: 64x64 hash kernel with / without the next-frame prefetch (static launches), product grid rule."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
fr = synth.frames_gray(10000, seed=2)
for nf in [int(x) for x in os.environ.get("NS", "1000,4000,8192,10000,14000,20000,40000").split(",")]:
    d_f = L.DeviceBuffer(nf * 4096)
    for r0 in range(0, nf, 10000):
        m = min(10000, nf - r0)
        L.check(lib.hvd_memcpy_h2d(C.c_void_p(d_f.ptr + r0 * 4096), fr.ctypes.data, m * 4096))
    d_h, d_q = L.DeviceBuffer(32 * nf), L.DeviceBuffer(4 * nf)
    out = []
    for pref in (0, 1, 0, 1):
        L.check(lib.hvd_debug_set(b"pdq_hash_prefetch", pref))
        ks = []
        R = int(os.environ.get("REPS", 200))
        for r in range(R):
            L.check(lib.hvd_timer_start())
            L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, nf, 64, 64, 1, None, d_h.ptr, d_q.ptr))
            ms = C.c_float(0)
            L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r >= R // 2:
                ks.append(ms.value)
        out.append(np.mean(ks) * 1e3)
    print(f"n={nf:7d}: off {out[0]:8.2f} {out[2]:8.2f} us   on {out[1]:8.2f} {out[3]:8.2f} us   -> {nf / out[3] * 1e6:.4g} frames/s", flush=True)
    d_f.free(); d_h.free(); d_q.free()
