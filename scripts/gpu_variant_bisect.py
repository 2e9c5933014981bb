"""Dev probe: run one all-pairs variant on the golden DB in this process (crash bisect)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M

v = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = L.init(0)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "hamming_db.npz"))
db = g["db"]; n = len(db)
d_db = L.DeviceBuffer.from_array(db)
d_img = M.expand_fp4(d_db.ptr, n)
cap = 1 << 16
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
for r in range(reps):
    d_cnt.zero()
    M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, v)
    cnt = int(d_cnt.to_array(np.uint64, 1)[0])
    got = M.merge_pairs([d_pairs.to_array(L.PAIR_DTYPE, cnt)])
    print("variant", v, "rep", r, "n", n, "pairs", cnt, "equal", np.array_equal(got, g["pairs"]), flush=True)
