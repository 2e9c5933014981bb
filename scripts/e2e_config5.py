"""BASELINE config 5 on one GPU: 50k synthetic videos x 64 frames.
Hashing half: 3.2M 64x64 frames hashed in HBM (a 10k-frame synthetic batch, 320 passes: the frame
content does not change the kernel's work). Search half: 50k videos x 64 synthetic frame hashes with
planted near-copies -> all video pairs (5.1e12 frame comparisons), checked against the planted set."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
lib = L.init(0)
V, F = 50_000, 64
fr = synth.frames_gray(10_000, seed=5)
d_f = L.DeviceBuffer.from_array(fr); d_h = L.DeviceBuffer(32 * len(fr)); d_q = L.DeviceBuffer(4 * len(fr))
passes = V * F // len(fr)
L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, len(fr), 64, 64, 1, None, d_h.ptr, d_q.ptr)); L.check(lib.hvd_dev_sync())
L.check(lib.hvd_timer_start())
for _ in range(passes):
    L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, len(fr), 64, 64, 1, None, d_h.ptr, d_q.ptr))
ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
print(f"hash: {V * F} frames in {ms.value:.1f} ms = {V * F / ms.value / 1e3:.1f} Mframes/s")
frames, offsets, planted = synth.video_hashes(V, seed=5, frames_per_video=F, copy_fraction=0.02, max_flips=24)
hvd_amd.match_videos(frames[: 64 * 1000], offsets[:1001], 31)  # warm-up
t = time.perf_counter()
recs = hvd_amd.match_videos(frames, offsets, 31)
dt = time.perf_counter() - t
pairs = hvd_amd.search.similar_video_pairs(recs, np.diff(offsets), 50.0)
got = {(int(a), int(b)) for a, b in pairs}
found = sum((int(s), int(d)) in got for s, d in planted)
ncmp = (V * F) * (V * F - 1) / 2
print(f"search: {V} videos, {ncmp:.3g} frame comparisons in {dt * 1e3:.1f} ms (host buffers in, video records out) "
      f"= {ncmp / dt / 1e12:.2f} Tcmp/s; {len(recs)} video records, {len(pairs)} pairs >= 50 %, "
      f"{found}/{len(planted)} planted copies among them")
