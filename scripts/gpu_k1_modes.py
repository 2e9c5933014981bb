"""Developer check: strict (VALU) vs fma (matrix-core) DCT mode of the frame hash: parity and timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
from oracle import oracle as O
lib = L.init(0)
fr = synth.frames_gray(10000, seed=2)
for mode in ("strict", "fma"):
    hvd_amd.vpdq.set_dct_mode(mode)
    ho, qo = O.hash_frames(fr, num_threads=16, fma=(mode == "fma"))
    h, q = hvd_amd.vpdq.hash_frames(fr)
    print(f"mode={mode}: hash mismatches {int((h != ho).any(1).sum())} quality mismatches {int((q != qo).sum())}")
    for n in (10_000, 400_000):
        f = np.concatenate([fr] * (n // len(fr)))
        d_f = L.DeviceBuffer.from_array(f); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
        best = 1e9
        for r in range(6):
            L.check(lib.hvd_timer_start())
            L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 64, 64, 1, None, d_h.ptr, d_q.ptr))
            ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
            if r: best = min(best, ms.value)
        print(f"   n={n}: {best:.3f} ms  {n / best / 1e3:.1f} Mframes/s")
hvd_amd.vpdq.set_dct_mode("strict")
