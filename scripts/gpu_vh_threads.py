"""Dev probe: hash_frame(bytes) at 512x512 RGB24 for several VideoHasher num_threads (host copy threads)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth, vpdq
L.init(0)
rgb = synth.frames_rgb(16, seed=6)
video = np.ascontiguousarray(rgb[np.arange(300) % 16])
hh, qq = vpdq.hash_frames(video)
want = hh[qq >= 31].tobytes()
frames = [video[k].tobytes() for k in range(300)]
import ctypes as C
lib = L.load()
for mode in (1, 0, 1, 0):
  L.check(lib.hvd_debug_set(b"copy_nt", mode))
  lv = C.c_int(0); lib.hvd_debug_get(b"copy_nt", C.byref(lv))
  print("copy_nt level", lv.value, flush=True)
  for nt in (1, 2, 3, 4, 6, 8):
    for rep in range(2):
        t = time.perf_counter()
        for v in range(10):
            hs = vpdq.VideoHasher(1, 512, 512, nt)
            for f in frames:
                hs.hash_frame(f)
            assert hs.finish().bytes == want
        dt = time.perf_counter() - t
    print("   ", end="")
    print(f"num_threads {nt}: {3000 / dt:8.0f} frames/s  {3000 * 786432 / dt / 1e9:6.2f} GB/s  {dt / 3000 * 1e6:6.2f} us per frame", flush=True)
