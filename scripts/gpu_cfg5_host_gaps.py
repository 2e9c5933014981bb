"""Dev probe: where the host-side milliseconds of one config-5 video search go (Python around hvd_dev_vpdq_match_videos)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hvd_amd import _lib as L, pipeline
from hvd_amd._lib import VMATCH_DTYPE
lib = L.init(0)
V, F = 50_000, 64
d_frames = L.DeviceBuffer(V * F * 4096)
L.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, V, F, 5, None))
_, recs0, lib5 = pipeline.dedupe_frames_on_device(d_frames.ptr, np.arange(V + 1, dtype=np.int64) * F, 64, 64, 1, keep_library=True)
d_frames.free()
for rep in range(3):
    T = [time.perf_counter()]
    d_out, d_cnt, _ = pipeline._record_buffers(max(4096, V)); T.append(time.perf_counter())
    img = lib5.image().ptr; T.append(time.perf_counter())
    L.check(lib.hvd_dev_vpdq_match_videos(img, lib5.n_frames, lib5.d_video.ptr, 31, 0, 1, d_out.ptr, max(4096, V), d_cnt.ptr)); T.append(time.perf_counter())
    cnt = int(d_cnt.to_array(np.uint64, 1)[0]); T.append(time.perf_counter())
    recs = d_out.to_array(VMATCH_DTYPE, cnt); T.append(time.perf_counter())
    recs = recs[np.argsort((recs["a"].astype(np.uint64) << np.uint64(32)) | recs["b"].astype(np.uint64), kind="stable")]; T.append(time.perf_counter())
    us = C.c_int(0); parts = []
    for key in (b"vmatch_us_local", b"vmatch_us_exchange", b"vmatch_us_fold"):
        L.check(lib.hvd_debug_get(key, C.byref(us))); parts.append(us.value / 1e3)
    names = ["buffers", "image", "C call", "count D2H", "records D2H", "sort"]
    print(" | ".join(f"{n} {1e3 * (T[i + 1] - T[i]):.3f}" for i, n in enumerate(names)), "| inside C: local/exchange/fold ms", parts, flush=True)
