"""Developer check: 512x512 rgb24 front-end parity + timing at saturation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
from oracle import oracle as O
lib = L.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base = synth.frames_rgb(16, seed=6)
ho, qo = O.hash_frames(base, num_threads=16)
fr = np.concatenate([base] * (n // 16))
sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(n, 512, 512, 3, C.byref(sb)))
d_f = L.DeviceBuffer.from_array(fr); d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
gray = synth.frames_gray(8, seed=9, h=512, w=512)
hg, qg = hvd_amd.vpdq.hash_frames(gray); hgo, qgo = O.hash_frames(gray, num_threads=8)
print("gray512 parity:", np.array_equal(hg, hgo), np.array_equal(qg, qgo))
for fused in (0, 1, 2, 3, 4):
  L.check(lib.hvd_debug_set(b"pdq_fused_down512", 1 if fused else 0))
  L.check(lib.hvd_debug_set(b"pdq_down512_systolic", 1 if fused == 2 else 0))
  L.check(lib.hvd_debug_set(b"pdq_down512_split_d", 1 if fused >= 3 else 0))
  L.check(lib.hvd_debug_set(b"pdq_down512_strip64", 1 if fused == 4 else 0))
  if fused >= 3:
      hg, qg = hvd_amd.vpdq.hash_frames(gray)
      print("gray512 split-D parity:", np.array_equal(hg, hgo), np.array_equal(qg, qgo))
  if fused == 2:
      hg, qg = hvd_amd.vpdq.hash_frames(gray)
      print("gray512 systolic parity:", np.array_equal(hg, hgo), np.array_equal(qg, qgo))
  best = 1e9
  for r in range(4):
    L.check(lib.hvd_timer_start())
    L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 512, 512, 3, d_s.ptr, d_h.ptr, d_q.ptr))
    ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
    if r: best = min(best, ms.value)
  h = d_h.to_array(np.uint8, 32 * n).reshape(-1, 32); q = d_q.to_array(np.int32, n)
  ok = np.array_equal(h, np.concatenate([ho] * (n // 16))) and np.array_equal(q, np.concatenate([qo] * (n // 16)))
  print(f"rgb512 fused={fused} n={n}: parity {ok}  {best:.3f} ms  {n / best:.1f} kframes/s  ({n * 786468 / best / 1e6:.1f} GB/s algorithmic)")
