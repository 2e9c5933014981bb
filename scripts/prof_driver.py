"""Minimal workload for rocprofv3 counter passes: the bench's dominant kernels, few launches.
  python scripts/prof_driver.py [n_hashes] [variant]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth, multigpu as M

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 13
lib = L.init(0)
db, _ = synth.hash_db(n, seed=3)
d_db = L.DeviceBuffer.from_array(db)
d_img = M.expand_fp4(d_db.ptr, n)
d_pairs = L.DeviceBuffer(16 << 20); d_cnt = L.DeviceBuffer(8)
for _ in range(3):
    d_cnt.zero()
    L.check(lib.hvd_dev_expand_fp4(d_db.ptr, n, d_img.ptr))
    M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, 1 << 20, d_cnt.ptr, variant)
    L.check(lib.hvd_dev_sync())
fr = synth.frames_gray(10000, seed=2)
d_f = L.DeviceBuffer.from_array(fr); d_h = L.DeviceBuffer(32 * len(fr)); d_q = L.DeviceBuffer(4 * len(fr))
for _ in range(3):
    L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, len(fr), 64, 64, 1, None, d_h.ptr, d_q.ptr))
L.check(lib.hvd_dev_sync())
print("prof_driver done", int(d_cnt.to_array(np.uint64, 1)[0]))
