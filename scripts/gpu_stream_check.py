"""Developer check: throughput of the legacy VideoHasher.hash_frame path (host frames -> hashes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
L.init(0)
for (w, h, ch, n) in ((512, 512, 3, 2000), (64, 64, 1, 50000)):
    base = synth.frames_rgb(16, seed=6) if ch == 3 else synth.frames_gray(1000, seed=2)
    frames = [bytes(base[i % len(base)]) for i in range(n)]
    for rep in range(2):
        t = time.perf_counter()
        hs = hvd_amd.VideoHasher(1, w, h, 0)
        for f in frames:
            hs.hash_frame(f)
        ph = hs.finish()
        dt = time.perf_counter() - t
    print(f"VideoHasher {w}x{h}x{ch}: {n} frames in {dt*1e3:.1f} ms = {n/dt/1e3:.1f} kframes/s "
          f"({n*w*h*ch/dt/1e9:.2f} GB/s host->device), kept {len(ph)}")
