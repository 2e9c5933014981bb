"""Dev probe: k_down512w at 6144 frames (wave kernel forced), 10 launches, for the library named by HVD_LIB_PATH."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
ch = int(os.environ.get("CH", "3"))
base = synth.frames_rgb(16, seed=6) if ch == 3 else synth.frames_gray(16, seed=6, h=512, w=512)
n = int(os.environ.get("N", "6144"))
fr = np.concatenate([base] * (n // 16))
d_f = L.DeviceBuffer.from_array(fr)
sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(n, 512, 512, ch, C.byref(sb)))
d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
L.check(lib.hvd_debug_set(b"pdq_down512_wave", 2))
L.check(lib.hvd_debug_set(b"pdq_down512_wave_grid", int(os.environ.get("WGRID", "0"))))
ts = []
for r in range(32):
    L.check(lib.hvd_timer_start())
    L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 512, 512, ch, d_s.ptr, d_h.ptr, d_q.ptr))
    ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
    if r >= 22: ts.append(ms.value)
print(" ".join(f"{t:.3f}" for t in ts))
print(f"{os.environ.get('HVD_LIB_PATH', 'default'):40s} ch={ch} n={n}: mean {np.mean(ts):.3f} ms  min {np.min(ts):.3f} ms  {n / np.mean(ts):.0f} kf/s  frac {n / np.mean(ts) * 1e3 * (512 * 512 * ch + 36) / 8e12:.3f}")
