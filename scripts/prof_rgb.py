"""Workload for rocprofv3 counter passes on the 512x512 front-end. argv[1]: pdq_down512_wave (0 workgroup kernel, 1 by batch size, 2 wave kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
L.check(lib.hvd_debug_set(b"pdq_down512_wave", int(sys.argv[1]) if len(sys.argv) > 1 else 1))
n = 6144
base = synth.frames_rgb(16, seed=6)
fr = np.concatenate([base] * (n // 16))
sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(n, 512, 512, 3, C.byref(sb)))
d_f = L.DeviceBuffer.from_array(fr); d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
for r in range(3):
    L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 512, 512, 3, d_s.ptr, d_h.ptr, d_q.ptr))
L.check(lib.hvd_dev_sync())
print("done")
