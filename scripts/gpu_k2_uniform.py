"""Dev probe: all-pairs forms on a uniform random DB (1 M hashes), kernel-only timing."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, synth
lib = L.init(0)
if os.environ.get("CHUNK"): L.check(lib.hvd_debug_set(b"mfma_col_chunk_max", int(os.environ["CHUNK"])))
n = int(os.environ.get("N", 1_000_000))
db, _ = synth.hash_db(n, seed=3)
d_db = L.DeviceBuffer.from_array(db)
d_img = M.expand_fp4(d_db.ptr, n)
cap = 1 << 20
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
for v in [int(x) for x in (sys.argv[1:] or ["9", "12", "13"])]:
    ks = []
    for r in range(10):
        d_cnt.zero()
        L.check(lib.hvd_timer_start())
        M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, v)
        ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
        if r >= 3: ks.append(ms.value)
    print(f"{os.environ.get('HVD_LIB_PATH', 'default'):32s} variant {v:2d}: {np.mean(ks):8.3f} ms  {n * (n - 1) / 2 / np.mean(ks) / 1e9:.2f} Tcmp/s  pairs {int(d_cnt.to_array(np.uint64, 1)[0])}", flush=True)
