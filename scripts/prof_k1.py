"""Workload for rocprofv3 counter passes on k_pdq_hash64. argv[1] = number of frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
fr = synth.frames_gray(10000, seed=2)
fr = np.concatenate([fr] * (n // 10000))
d_f = L.DeviceBuffer.from_array(fr); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
for r in range(3):
    L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 64, 64, 1, None, d_h.ptr, d_q.ptr))
L.check(lib.hvd_dev_sync())
print("done")
