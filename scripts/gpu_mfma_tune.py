"""Developer check: sweep of the MFMA kernel's column-chunk cap at 1M hashes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth, multigpu as M
lib = L.init(0)
n = 1_000_000
db, _ = synth.hash_db(n, seed=3)
d_db = L.DeviceBuffer.from_array(db)
d_pairs = L.DeviceBuffer(16 << 20); d_cnt = L.DeviceBuffer(8)
from oracle import oracle as O
small, _ = synth.hash_db(3000, seed=9, plant_fraction=0.05)
want = O.allpairs(small, 31)
for chunk in (1, 2, 4, 6, 2):
    L.check(lib.hvd_debug_set(b"fp4_code", chunk))
    print("fp4_code", chunk, "parity:", np.array_equal(hvd_amd.allpairs_hamming(small, 31), want))
    d_img = M.expand_fp4(d_db.ptr, n)
    for v in (9, 8):
        best = 1e9
        for r in range(4):
            d_cnt.zero()
            L.check(lib.hvd_timer_start())
            M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, 1 << 20, d_cnt.ptr, v)
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r: best = min(best, ms.value)
        print(f"chunk_max={chunk} variant={v}: {best:.2f} ms {n*(n-1)/2/best/1e9:.2f} Tcmp/s pairs={int(d_cnt.to_array(np.uint64,1)[0])}")
