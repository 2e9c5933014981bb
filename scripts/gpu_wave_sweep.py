"""Developer check: k_down512 (workgroup per frame) vs k_down512w (wave per frame) across batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
base = synth.frames_rgb(16, seed=6)
nmax = 6144
fr = np.concatenate([base] * (nmax // 16))
d_f = L.DeviceBuffer.from_array(fr)
sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(nmax, 512, 512, 3, C.byref(sb)))
d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32 * nmax); d_q = L.DeviceBuffer(4 * nmax)
for n in (16, 64, 128, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144):
    row = []
    for wave in (0, 2):
        L.check(lib.hvd_debug_set(b"pdq_down512_wave", wave))
        best = 1e9
        for r in range(5):
            L.check(lib.hvd_timer_start())
            L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 512, 512, 3, d_s.ptr, d_h.ptr, d_q.ptr))
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r: best = min(best, ms.value)
        row.append(best)
    print(f"n={n:5d}: workgroup/frame {row[0]:8.3f} ms ({n / row[0]:7.1f} kf/s)   wave/frame {row[1]:8.3f} ms ({n / row[1]:7.1f} kf/s)", flush=True)
