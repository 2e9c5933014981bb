"""Developer check: 512x512 GRAY frames, workgroup kernel vs wave kernel across batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
base = synth.frames_gray(16, seed=9, h=512, w=512)
nmax = 8192
d_f = L.DeviceBuffer(nmax * 262144)
for rep in range(nmax // 16):
    L.check(lib.hvd_memcpy_h2d(C.c_void_p(d_f.ptr + rep * base.nbytes), base.ctypes.data, base.nbytes))
sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(nmax, 512, 512, 1, C.byref(sb)))
d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32 * nmax); d_q = L.DeviceBuffer(4 * nmax)
for n in (256, 1024, 4096, 8192):
    row = []
    for wave in (0, 2):
        L.check(lib.hvd_debug_set(b"pdq_down512_wave", wave))
        best = 1e9
        for r in range(5):
            L.check(lib.hvd_timer_start())
            L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 512, 512, 1, d_s.ptr, d_h.ptr, d_q.ptr))
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r: best = min(best, ms.value)
        row.append(best)
    print(f"gray n={n:5d}: workgroup/frame {row[0]:8.3f} ms ({n / row[0]:7.1f} kf/s)   wave/frame {row[1]:8.3f} ms ({n / row[1]:7.1f} kf/s)", flush=True)
