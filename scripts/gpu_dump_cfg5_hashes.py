"""Dev: dump the kept frame hashes of the first 3000 videos of the config-5 generator (for offline survivor statistics) and
print what the probe counts on the full library."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hvd_amd import _lib as L, pipeline
lib = L.init(0)
V, F = 50_000, 64
rng = np.random.default_rng(5)
copy_of = np.full(V, -1, dtype=np.int32)
m = int(round(V * 0.02))
dst = rng.choice(np.arange(1, V), size=m, replace=False)
is_dst = np.zeros(V, dtype=bool); is_dst[dst] = True
copy_of[dst] = rng.choice(np.flatnonzero(~is_dst), size=m)
d_copy = L.DeviceBuffer.from_array(copy_of)
d_frames = L.DeviceBuffer(V * F * 4096)
L.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, V, F, 5, d_copy.ptr))
raw_off = np.arange(V + 1, dtype=np.int64) * F
_, recs, lib5 = pipeline.dedupe_frames_on_device(d_frames.ptr, raw_off, 64, 64, 1, keep_library=True)
for k in (b"mfma_auto_form", b"mfma_probe_survivors", b"mfma_probe_survivors_hi", b"mfma_auto_half"):
    v = C.c_int(0); L.check(lib.hvd_debug_get(k, C.byref(v))); print(k.decode(), v.value)
off = lib5.offsets()
h = lib5.hashes()
n = int(off[3000])
os.makedirs("gpurun_out/r05g", exist_ok=True)
np.savez_compressed("gpurun_out/r05g/cfg5_hashes.npz", hashes=h[:n], offsets=off[:3001], copy_of=copy_of[:3000])
print("kept", lib5.n_frames, "dumped", n, "records", len(recs))
