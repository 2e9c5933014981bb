import sys, os
sys.path.insert(0, ".")
import ctypes as C, numpy as np, hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
fr = synth.frames_gray(10000, seed=2)
d_f = L.DeviceBuffer.from_array(fr); d_h, d_q = L.DeviceBuffer(32 * 10000), L.DeviceBuffer(4 * 10000)
for src in (0, 2):
    L.check(lib.hvd_debug_set(b"pdq_dct_from_lds", src))
    for grid in (0, 2500, 2304, 2048, 1792, 1536, 1280, 1250, 1024, 834):
        L.check(lib.hvd_debug_set(b"pdq_hash_grid", grid))
        ks = []
        for r in range(30):
            L.check(lib.hvd_timer_start()); L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, 10000, 64, 64, 1, None, d_h.ptr, d_q.ptr)); ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r >= 5: ks.append(ms.value)
        print(f"operand {src} grid {grid:5d}: {np.mean(ks)*1e3:7.2f} us  {1e4/np.mean(ks)*1e3:.4g} frames/s", flush=True)
