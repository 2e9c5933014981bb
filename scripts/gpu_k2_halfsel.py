"""Dev probe: all-pairs on a DB whose bits 0..127 barely separate hashes (probe -> first stage on bits 128..255)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, synth
lib = L.init(0)
n = 1_000_000
rng = np.random.default_rng(5)
db, _ = synth.hash_db(n, seed=3)
common = rng.integers(0, 256, 16, dtype=np.uint8)
flips = rng.integers(0, 256, (n, 16), dtype=np.uint8) & rng.integers(0, 256, (n, 16), dtype=np.uint8) & rng.integers(0, 256, (n, 16), dtype=np.uint8) & rng.integers(0, 256, (n, 16), dtype=np.uint8)
db[:, :16] = common[None, :] ^ flips  # ~8 of 128 bits flipped per hash
d_db = L.DeviceBuffer.from_array(db)
d_img = M.expand_fp4(d_db.ptr, n)
cap = 1 << 20
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
for v in (13, 12, 9):
    ks = []
    for r in range(3 if v != 13 else 6):
        d_cnt.zero()
        L.check(lib.hvd_timer_start())
        M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, v)
        ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
        if r >= 1: ks.append(ms.value)
    h = C.c_int(0); f = C.c_int(0)
    L.check(lib.hvd_debug_get(b"mfma_auto_half", C.byref(h))); L.check(lib.hvd_debug_get(b"mfma_auto_form", C.byref(f)))
    print(f"variant {v:2d}: {np.mean(ks):9.2f} ms  pairs {int(d_cnt.to_array(np.uint64, 1)[0])}  (last probe: form {f.value} half {h.value})", flush=True)
