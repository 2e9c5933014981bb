"""Dev: hash_frame(bytes), 512x512 RGB24, 300-frame videos: copy threads 2 / 4 / 6 / 8 interleaved on ONE box, several rounds (single
rounds scatter by 10-20 %: only an interleaved A/B says anything)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hvd_amd import _lib as L, synth, vpdq
lib = L.init(0)
F = 300
distinct = synth.frames_rgb(16, seed=6)
video = np.ascontiguousarray(distinct[np.arange(F) % 16])
as_bytes = [video[k].tobytes() for k in range(F)]
fb = len(as_bytes[0])
def run(nt):
    hs = vpdq.VideoHasher(1, 512, 512, nt)
    for f in as_bytes:
        hs.hash_frame(f)
    return hs.finish()
res = {nt: [] for nt in (2, 4, 6, 8)}
for nt in res:
    run(nt)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for nt in res:
        run(nt)
        t = time.perf_counter()
        for _ in range(10):
            run(nt)
        res[nt].append((time.perf_counter() - t) / 10 / F * 1e6)
for nt, v in res.items():
    s = sorted(v)
    print(f"threads {nt}: median {s[len(s)//2]:6.2f} us/frame  min {s[0]:6.2f}  max {s[-1]:6.2f}   rounds {[round(x,1) for x in v]}")
print("cpus usable:", len(os.sched_getaffinity(0)), "| quota:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
