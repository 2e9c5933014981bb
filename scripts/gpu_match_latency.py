"""Developer check: latency of the legacy one-pair entry (matchHashBytes), as the VP-tree would call it."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, synth
L.init(0)
fr, off, _ = synth.video_hashes(200, seed=1, frames_per_video=64, copy_fraction=0.1)
blobs = [fr[off[v]:off[v + 1]].tobytes() for v in range(200)]
hvd_amd.matchHashBytes(blobs[0], blobs[1], 31)
t = time.perf_counter(); n = 0
for a in range(60):
    for b in range(60):
        hvd_amd.calculate_distance(blobs[a], blobs[b]); n += 1
dt = time.perf_counter() - t
print(f"calculate_distance (64x64 frame hashes): {dt / n * 1e6:.1f} us per call ({n} calls)")
