"""Dev probe: all-pairs kernel timing for a list of variants (A/B of builds via HVD_LIB_PATH)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, synth

lib = L.init(0)
n = int(os.environ.get("N", 1_000_000))
variants = [int(v) for v in (sys.argv[1:] or ["9", "12", "13"])]
db, _ = synth.hash_db(n, seed=3)
d_db = L.DeviceBuffer.from_array(db)
d_img = M.expand_fp4(d_db.ptr, n)
cap = 1 << 20
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
for rnd in range(int(os.environ.get("ROUNDS", 2))):
    for v in variants:
        ks = []
        for r in range(8):
            d_cnt.zero()
            L.check(lib.hvd_timer_start())
            M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, v)
            ms = C.c_float(0)
            L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r >= 2:
                ks.append(ms.value)
        cnt = int(d_cnt.to_array(np.uint64, 1)[0])
        print(f"{os.path.basename(L.LIB_PATH):24s} variant {v:2d}: {np.mean(ks):8.3f} ms +- {np.std(ks):.3f}  pairs {cnt}", flush=True)
