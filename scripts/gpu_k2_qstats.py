"""Dev probe (HVD_K2_QSTATS build): routes taken by the pair-queue form's surviving tiles on the config-5 frame hashes."""
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
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
def stat(k):
    v = C.c_int(0); L.check(lib.hvd_debug_get(b"mfma_qstat%d" % k, C.byref(v))); return v.value
for k in range(16): stat(k)
d_cnt.zero()
M.launch_allpairs(lib, libr.d_hashes.ptr, img.ptr, nk, libr.d_video.ptr, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, int(sys.argv[1]) if len(sys.argv) > 1 else 15)
L.check(lib.hvd_dev_sync())
s = [stat(k) for k in range(16)]
tiles = nk * (nk - 1) / 2 / 1024
print("kept", nk, "tiles %.3g" % tiles)
print("surviving tiles", s[0], "= %.3f of tiles; survivors %d = %.3g of pairs; per surviving tile %.2f" % (s[0] / tiles, s[3], s[3] / (tiles * 1024), s[3] / max(s[0], 1)))
print("tile route: lanes>16:", s[1], " heavy lane(>2):", s[2])
print("survivors per surviving tile histogram (<=1,2,4,8,16,32,64,more):", s[8:16])
print("settlements", s[4], "entries settled", s[5], "= %.1f each; at the end of a workgroup: %d; forced by one full queue: %d" % (s[5] / max(s[4], 1), s[6], s[7]))
