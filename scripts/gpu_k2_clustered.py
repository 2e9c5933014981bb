"""Dev probe: explicit all-pairs forms on bench.py's clustered hash DBs (1 M hashes, clusters of near-identical hashes
scattered over the DB: every hit is an isolated true pair)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, synth
lib = L.init(0)
n = 1_000_000
cap = 1 << 23
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
for ncl, csz in ((10_000, 10), (1_000, 100)):
    db, _ = synth.hash_db_clustered(n, ncl, csz, seed=8)
    d_db = L.DeviceBuffer.from_array(db)
    d_img = M.expand_fp4(d_db.ptr, n)
    for v in [int(x) for x in (sys.argv[1:] or ["9", "15", "18", "12", "13"])]:
        ks = []
        for r in range(4):
            d_cnt.zero()
            L.check(lib.hvd_timer_start())
            M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, v)
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r: ks.append(ms.value)
        print(f"{ncl} clusters of {csz}: variant {v:2d}: {np.mean(ks):8.2f} ms  pairs {int(d_cnt.to_array(np.uint64, 1)[0])}", flush=True)
    d_db.free(); d_img.free()
