"""Dev probe: where does the host side of hash_frame(bytes) spend its time (copy / submit / wait for a slot)?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hvd_amd import _lib as L, synth, vpdq
lib = L.init(0)
rgb = synth.frames_rgb(16, seed=6)
video = np.ascontiguousarray(rgb[np.zeros(300, dtype=int)])  # ONE repeated frame: acquire_only is valid for any batch layout
hh, qq = vpdq.hash_frames(video)
want = hh[qq >= 31].tobytes()
frames = [video[k].tobytes() for k in range(300)]


def take():
    out = []
    for k in (b"hasher_us_copy", b"hasher_us_submit", b"hasher_us_wait"):
        v = C.c_int(0); L.check(lib.hvd_debug_get(k, C.byref(v))); out.append(v.value)
    return out


for batch_mb in ((64,) if os.environ.get('HVD_LIB_PATH') else (32, 64, 16)):
    for nt in (1, 4, 8):
        for feed in ("bytes", "acquire_only"):
            for rep in range(3):
                take()
                use = "acquire_copy" if (feed == "acquire_only" and rep == 0) else feed  # (rep 0 fills the slots for acquire_only)
                t = time.perf_counter()
                for v in range(10):
                    hs = vpdq.VideoHasher(1, 512, 512, nt, batch_bytes=batch_mb << 20)
                    if use == "bytes":
                        for f in frames:
                            hs.hash_frame(f)
                    elif use == "acquire_copy":
                        for k in range(300):
                            np.copyto(hs.acquire_frame(3), video[k]); hs.commit_frame()
                    else:
                        for k in range(300):
                            hs.acquire_frame(3); hs.commit_frame()
                    t1 = time.perf_counter()
                    got = hs.finish()
                    assert got.bytes == want or os.environ.get('HVD_LIB_PATH')
                dt = time.perf_counter() - t
                c, s, w = take()
            print(f"batch {batch_mb:3d} MiB threads {nt} {feed:12s}: {dt / 3000 * 1e6:6.2f} us/frame = {3000 * 786432 / dt / 1e9:5.1f} GB/s | per frame: copy {c / 3000:5.2f} submit {s / 3000:5.2f} "
                  f"wait {w / 3000:5.2f} other {dt / 3000 * 1e6 - (c + s + w) / 3000:5.2f} us", flush=True)
