"""Developer check: dump the 64x64 down-sampled luma of one 512x512 frame and compare with the numpy restatement."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
from oracle import pdq_numpy as P
lib = L.init(0)
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fr = synth.frames_gray(1, seed=9, h=512, w=512) if ch == 1 else synth.frames_rgb(1, seed=6)
want = P.jarosz_decimate(P.luma_gray(fr[0]) if ch == 1 else P.luma_rgb(fr[0]))
sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(1, 512, 512, ch, C.byref(sb)))
d_f = L.DeviceBuffer.from_array(fr); d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32); d_q = L.DeviceBuffer(4)
L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, 1, 512, 512, ch, d_s.ptr, d_h.ptr, d_q.ptr)); L.check(lib.hvd_dev_sync())
got = d_s.to_array(np.float32, 4096).reshape(64, 64)
bad = got.view(np.uint32) != want.view(np.uint32)
print("channels", ch, "mismatching outputs:", int(bad.sum()))
print("bad rows i:", np.nonzero(bad.any(1))[0].tolist())
print("bad cols j:", np.nonzero(bad.any(0))[0].tolist())
ii, jj = np.nonzero(bad)
for k in range(min(8, len(ii))):
    print(ii[k], jj[k], got[ii[k], jj[k]], want[ii[k], jj[k]])
