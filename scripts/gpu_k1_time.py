"""Dev probe: 64x64 hash kernel timing, strict and fma, for the three DCT-operand sources."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, numpy as np, hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
fr = synth.frames_gray(10000, seed=2)
for nf in (10000, 400000):
    d_f = L.DeviceBuffer(nf * 4096)
    for r0 in range(0, nf, 10000):
        L.check(lib.hvd_memcpy_h2d(C.c_void_p(d_f.ptr + r0 * 4096), fr.ctypes.data, 10000 * 4096))
    d_h, d_q = L.DeviceBuffer(32 * nf), L.DeviceBuffer(4 * nf)
    ref = None
    for src in (0, 1, 2):
        L.check(lib.hvd_debug_set(b"pdq_dct_from_lds", src))
        ks = []
        for r in range(12):
            L.check(lib.hvd_timer_start())
            L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, nf, 64, 64, 1, None, d_h.ptr, d_q.ptr))
            ms = C.c_float(0)
            L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r >= 2:
                ks.append(ms.value)
        h = d_h.to_array(np.uint8, 32 * nf)
        ref = h if ref is None else ref
        print(f"n={nf} dct operand source {src}: {np.mean(ks):.4f} ms  {nf / np.mean(ks) * 1e3:.4g} frames/s  same hashes {np.array_equal(h, ref)}", flush=True)
    L.check(lib.hvd_debug_set(b"pdq_dct_from_lds", 3))
    d_f.free()
