"""Dev: does the kind of SOURCE memory matter to hash_frame()? (bench leg `bytes` 0.74 vs `buffer` 0.85 on one box.)
300-frame 512x512 RGB24 videos, one hasher per video, same native path; only where the frame bytes live differs."""
import os, sys, time, mmap
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hvd_amd import _lib as L, synth, vpdq
lib = L.init(0)
F = 300
distinct = synth.frames_rgb(16, seed=6)
video = np.ascontiguousarray(distinct[np.arange(F) % 16])
rows = video.reshape(F, -1)
fb = rows.shape[1]
as_bytes = [video[k].tobytes() for k in range(F)]
big = video.tobytes()
mv = memoryview(big)
views = [mv[k * fb:(k + 1) * fb] for k in range(F)]
same = [as_bytes[0]] * F
few = [as_bytes[k % 4] for k in range(F)]
# bytes objects that sit in ONE arena (bytearray slices copy; use a big bytearray + memoryview slices, writable)
ba = bytearray(big); mvb = memoryview(ba)
ba_views = [mvb[k * fb:(k + 1) * fb] for k in range(F)]
# page-aligned anonymous mmap with MADV_HUGEPAGE
mm = mmap.mmap(-1, F * fb + (2 << 20))
try:
    mm.madvise(mmap.MADV_HUGEPAGE)
except Exception as e:
    print("madvise:", e)
mm[:F * fb] = big
mmv = memoryview(mm)
mm_views = [mmv[k * fb:(k + 1) * fb] for k in range(F)]
srcs = {"bytes objects (bench `bytes`)": as_bytes, "array rows (bench `buffer`)": [rows[k] for k in range(F)],
        "memoryviews of one bytes": views, "memoryviews of one bytearray": ba_views, "mmap + MADV_HUGEPAGE views": mm_views,
        "one bytes object x300": same, "4 bytes objects cycled": few}
def run(src, nt=0):
    hs = vpdq.VideoHasher(1, 512, 512, nt)
    for f in src:
        hs.hash_frame(f)
    return hs.finish()
for rnd in range(2):
    for name, src in srcs.items():
        run(src)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.2:
            run(src)
        t = time.perf_counter()
        for _ in range(12):
            run(src)
        dt = (time.perf_counter() - t) / 12 / F
        print(f"round {rnd}  {name:34s} {dt * 1e6:6.2f} us/frame  {fb / dt / 1e9:5.1f} GB/s", flush=True)
for nt in (1, 2, 4, 8, 12):
    for name in ("bytes objects (bench `bytes`)", "array rows (bench `buffer`)"):
        src = srcs[name]
        run(src, nt); run(src, nt)
        t = time.perf_counter()
        for _ in range(8):
            run(src, nt)
        dt = (time.perf_counter() - t) / 8 / F
        print(f"threads {nt:2d}  {name:34s} {dt * 1e6:6.2f} us/frame  {fb / dt / 1e9:5.1f} GB/s", flush=True)
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| cpu:", [l for l in open("/proc/cpuinfo") if "model name" in l][0].strip())
