"""Dev probe: register form (12) against the queue forms (15 group masks, 18 panel marks) as a function of the first-stage survivor density:
the lower 128 bits of every hash are one of P prototypes (1024 / P survivors per 1024-pair tile), the upper 128 bits random."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M
lib = L.init(0)
n = int(os.environ.get("N", 400_000))
rng = np.random.default_rng(11)
cap = 1 << 20
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
for P in [int(x) for x in os.environ.get("PS", "16384,8192,4096,2048,1024,512,256").split(",")]:
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    db[:, :16] = rng.integers(0, 256, (P, 16), dtype=np.uint8)[rng.integers(0, P, n)]
    d_db = L.DeviceBuffer.from_array(db)
    d_img = M.expand_fp4(d_db.ptr, n)
    res = {}
    for v in (12, 15, 18, 13):
        ks = []
        for r in range(4):
            d_cnt.zero()
            L.check(lib.hvd_timer_start())
            M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, v)
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r: ks.append(ms.value)
        res[v] = (np.mean(ks), int(d_cnt.to_array(np.uint64, 1)[0]))
    print(f"P={P:6d} survivors/tile {1024 / P:6.3f}: form 12 {res[12][0]:8.3f} ms  form 15 {res[15][0]:8.3f} ms  form 18 {res[18][0]:8.3f} ms  ratio 15/12 {res[15][0] / res[12][0]:.3f}  18/12 {res[18][0] / res[12][0]:.3f}  auto {res[13][0]:8.3f} ms  pairs {res[12][1]} {res[15][1]} {res[18][1]} {res[13][1]}", flush=True)
    d_db.free(); d_img.free()
