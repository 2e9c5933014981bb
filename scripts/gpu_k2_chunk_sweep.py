"""Dev probe: column-chunk size of the all-pairs launch (hvd_debug_set mfma_col_chunk_max), variant 9, 1M uniform hashes."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, synth
lib = L.init(0)
n = 1_000_000
db, _ = synth.hash_db(n, seed=3)
d_db = L.DeviceBuffer.from_array(db); d_img = M.expand_fp4(d_db.ptr, n)
cap = 1 << 20
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
for rnd in range(2):
    for chunk in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        L.check(lib.hvd_debug_set(b"mfma_col_chunk_max", chunk))
        ks = []
        for r in range(6):
            d_cnt.zero(); L.check(lib.hvd_timer_start())
            M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, 9)
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r >= 1: ks.append(ms.value)
        print(f"col_chunk_max {chunk:6d}: {np.mean(ks):7.3f} ms +- {np.std(ks):.3f}", flush=True)
