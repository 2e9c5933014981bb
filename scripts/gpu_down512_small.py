"""Dev: latency of the 512x512 RGB24 front-end + hash for SMALL batches (what the tail of a streamed video waits for):
fused workgroup kernel (default below 704 frames) | wave-per-frame kernel | generic 4-launch path. ms, mean of 20."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from hvd_amd import _lib as L, synth
lib = L.init(0)
base = synth.frames_rgb(16, seed=6)
for n in (1, 5, 10, 21, 42, 84, 168, 256, 336, 512, 672):
    fr = np.concatenate([base] * ((n + 15) // 16))[:n]
    d_f = L.DeviceBuffer.from_array(fr)
    sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(n, 512, 512, 3, C.byref(sb)))
    d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
    out = {}
    for name, (fused, wave, strip) in {"workgroup": (1, 0, 32), "wg64": (1, 0, 64), "wave": (1, 2, 0), "generic": (0, 1, 0)}.items():
        L.check(lib.hvd_debug_set(b"pdq_fused_down512", fused)); L.check(lib.hvd_debug_set(b"pdq_down512_wave", wave))
        L.check(lib.hvd_debug_set(b"pdq_down512_strip", strip))
        ks = []
        for r in range(25):
            L.check(lib.hvd_timer_start())
            L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 512, 512, 3, d_s.ptr, d_h.ptr, d_q.ptr))
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r >= 5: ks.append(ms.value)
        out[name] = (float(np.mean(ks)), d_h.to_array(np.uint8, 32 * n).copy())
    L.check(lib.hvd_debug_set(b"pdq_fused_down512", 1)); L.check(lib.hvd_debug_set(b"pdq_down512_wave", 1)); L.check(lib.hvd_debug_set(b"pdq_down512_strip", 0))
    same = all(np.array_equal(out["workgroup"][1], out[k][1]) for k in out)
    print(f"n={n:4d}: workgroup {out['workgroup'][0]:.3f}  wg64 {out['wg64'][0]:.3f}  wave {out['wave'][0]:.3f}  generic {out['generic'][0]:.3f} ms   identical {same}", flush=True)
    for b in (d_f, d_s, d_h, d_q): b.free()
