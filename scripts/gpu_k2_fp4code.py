"""Dev probe: does the e2m1 magnitude of the +-v image (0.5 / 1 / 2 / 4: all exact) change the power-limited clock?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, synth
lib = L.init(0)
n = 1_000_000
db, _ = synth.hash_db(n, seed=3)
d_db = L.DeviceBuffer.from_array(db)
cap = 1 << 20
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
for rnd in range(2):
    for code in (2, 1, 4, 6):
        L.check(lib.hvd_debug_set(b"fp4_code", code))
        d_img = M.expand_fp4(d_db.ptr, n)
        ks = []
        for r in range(12):
            d_cnt.zero()
            L.check(lib.hvd_timer_start())
            M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, 9)
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r >= 4: ks.append(ms.value)
        print(f"fp4 code {code} (|v| = {{1: 0.5, 2: 1.0, 4: 2.0, 6: 4.0}}[code]): {np.mean(ks):8.3f} ms  pairs {int(d_cnt.to_array(np.uint64, 1)[0])}".replace("{{1: 0.5, 2: 1.0, 4: 2.0, 6: 4.0}}[code]", str({1: 0.5, 2: 1.0, 4: 2.0, 6: 4.0}[code])), flush=True)
        d_img.free()
L.check(lib.hvd_debug_set(b"fp4_code", 2))
