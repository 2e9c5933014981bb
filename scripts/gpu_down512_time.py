"""Dev probe: k_down512w timing (rgb + gray), mean of 10, and correctness vs the workgroup kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
for ch in (3, 1):
    base = synth.frames_rgb(16, seed=6) if ch == 3 else synth.frames_gray(16, 6, 512, 512)
    n = 6144 if ch == 3 else 8192
    fr = np.concatenate([base] * (n // 16))
    d_f = L.DeviceBuffer.from_array(fr)
    sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(n, 512, 512, ch, C.byref(sb)))
    d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
    res = {}
    for wave in (0, 2):
        L.check(lib.hvd_debug_set(b"pdq_down512_wave", wave))
        ks = []
        for r in range(12):
            L.check(lib.hvd_timer_start())
            L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 512, 512, ch, d_s.ptr, d_h.ptr, d_q.ptr))
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r >= 2: ks.append(ms.value)
        res[wave] = (np.mean(ks), np.std(ks), d_h.to_array(np.uint8, 32 * n).copy(), d_q.to_array(np.int32, n).copy())
    same = np.array_equal(res[0][2], res[2][2]) and np.array_equal(res[0][3], res[2][3])
    fb = 512 * 512 * ch + 36
    print(f"ch={ch} n={n}: wave kernel {res[2][0]:.3f} +- {res[2][1]:.3f} ms = {n / res[2][0] * 1e3:.4g} frames/s = "
          f"{n / res[2][0] * 1e3 * fb / 8e12:.3f} of HBM;  workgroup kernel {res[0][0]:.3f} ms; identical {same}", flush=True)
    L.check(lib.hvd_debug_set(b"pdq_down512_wave", 1))
    for b in (d_f, d_s, d_h, d_q): b.free()
