"""Stage timings of the chained config-5 pipeline on one GPU (dev probe)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, pipeline, search

lib = L.init(0)
if os.environ.get("CHUNK"): L.check(lib.hvd_debug_set(b"mfma_col_chunk_max", int(os.environ["CHUNK"])))
V, F = int(os.environ.get("V", 50000)), 64
n = V * F
d_frames = L.DeviceBuffer(n * 4096)
L.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, V, F, 5, None))
L.check(lib.hvd_dev_sync())
raw_off = np.arange(V + 1, dtype=np.int64) * F

def T(label, t0):
    L.check(lib.hvd_device_synchronize())
    t = time.perf_counter()
    print(f"{label:28s} {1e3 * (t - t0):9.2f} ms", flush=True)
    return t

for rep in range(2):
    print("--- rep", rep)
    t = time.perf_counter()
    d_h, d_q = pipeline.hash_frames_on_device(d_frames.ptr, n, 64, 64, 1)
    t = T("hash", t)
    libr = pipeline.DeviceLibrary.from_raw_hashes(d_h.ptr, d_q.ptr, n, raw_off)
    t = T("compact", t)
    libr.image()
    t = T("fp4 image", t)
    recs = libr.match_videos()
    t = T("match_videos", t)
    lens = libr.lengths()
    pairs = search.similar_video_pairs(recs, lens, 50.0)
    t = T("predicate", t)
    print("kept", libr.n_frames, "records", len(recs), "pairs", len(pairs))
    d_h.free(); d_q.free(); libr.free()
