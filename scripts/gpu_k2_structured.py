"""Dev probe: all-pairs forms on STRUCTURED frame hashes (config-5 generator), video mode off (frame pairs, group = video)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, pipeline
lib = L.init(0)
V, F = int(os.environ.get("V", 16000)), 64
n = V * F
d_frames = L.DeviceBuffer(n * 4096)
L.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, V, F, 5, None))
d_h, d_q = pipeline.hash_frames_on_device(d_frames.ptr, n, 64, 64, 1)
libr = pipeline.DeviceLibrary.from_raw_hashes(d_h.ptr, d_q.ptr, n, np.arange(V + 1, dtype=np.int64) * F)
d_frames.free()
nk = libr.n_frames
img = libr.image()
cap = 1 << 22
if os.environ.get("CHUNK"): L.check(lib.hvd_debug_set(b"mfma_col_chunk_max", int(os.environ["CHUNK"])))
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
print("kept frames", nk, "comparisons %.3g" % (nk * (nk - 1) / 2))
for v in [int(x) for x in (sys.argv[1:] or ["12", "18", "8", "13"])]:
    ks = []
    for r in range(4):
        d_cnt.zero()
        L.check(lib.hvd_timer_start())
        M.launch_allpairs(lib, libr.d_hashes.ptr, img.ptr, nk, libr.d_video.ptr, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, v)
        ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
        if r: ks.append(ms.value)
    cnt = int(d_cnt.to_array(np.uint64, 1)[0])
    print(f"variant {v:2d}: {np.mean(ks):8.2f} ms  {nk * (nk - 1) / 2 / np.mean(ks) / 1e9:.2f} Tcmp/s  pairs {cnt}", flush=True)
