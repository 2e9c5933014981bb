"""Dev probe: per-call cost of calculate_distance / match_counts with the resident match server on and off."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth, vpdq
lib = L.init(0)
vf, voff, _ = synth.video_hashes(33, seed=1, frames_per_video=64, copy_fraction=0.1)
blobs = [vf[voff[v]:voff[v + 1]].tobytes() for v in range(33)]
for mode in (1, 0, 1, 0):
    L.check(lib.hvd_debug_set(b"match_server", mode))
    hvd_amd.calculate_distance(blobs[0], blobs[1])
    for fn, name in ((hvd_amd.calculate_distance, "calculate_distance"), (lambda a, b: vpdq.match_counts(a, b, 31), "match_counts")):
        t = time.perf_counter(); n = 0
        for rep in range(4):
            for a in range(32):
                for b in range(32):
                    fn(blobs[a], blobs[b + 1]); n += 1
        print(f"match_server {mode} {name:20s}: {(time.perf_counter() - t) / n * 1e6:6.2f} us per call", flush=True)
    na = np.frombuffer(blobs[0], np.uint8); nb = np.frombuffer(blobs[1], np.uint8)
    q, t_ = C.c_int32(0), C.c_int32(0)
    t = time.perf_counter()
    for _ in range(4000):
        lib.hvd_match_two(blobs[0], 64, blobs[1], 64, 31, C.byref(q), C.byref(t_))
    print(f"match_server {mode} raw hvd_match_two       : {(time.perf_counter() - t) / 4000 * 1e6:6.2f} us per call")
if os.environ.get("HVD_LIB_PATH"):
    # timing build (-DHVD_MATCH_SERVER_TIMING): hdr[8] = 10 ns ticks from "request seen" to "answer ready" inside the server
    L.check(lib.hvd_debug_set(b"match_server", 1))
    q, t_ = C.c_int32(0), C.c_int32(0)
    import ctypes
    tk = []
    for _ in range(2000):
        lib.hvd_match_two(blobs[0], 64, blobs[1], 64, 31, C.byref(q), C.byref(t_))
        v = C.c_int(0); lib.hvd_debug_get(b"match_server_ticks", C.byref(v)); tk.append(v.value)
    print("in-kernel us (seen -> answer ready): median", np.median(tk) / 100.0, "min", min(tk) / 100.0)
