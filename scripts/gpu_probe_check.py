import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, synth
lib = L.init(0)
n = 1_000_000
cap = 1 << 23
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
def dbg(k):
    v = C.c_int(0); L.check(lib.hvd_debug_get(k, C.byref(v))); return v.value
for ncl, csz in ((10_000, 10), (1_000, 100)):
    db, _ = synth.hash_db_clustered(n, ncl, csz, seed=8)
    d_db = L.DeviceBuffer.from_array(db); d_img = M.expand_fp4(d_db.ptr, n)
    d_cnt.zero()
    M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, 13)
    L.check(lib.hvd_dev_sync())
    print(ncl, csz, "form", dbg(b"mfma_auto_form"), "survivors", dbg(b"mfma_probe_survivors"))
    d_db.free(); d_img.free()
