"""Dev probe: which pairs does variant V miss against variant 12 (uniform DB with planted pairs)?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hvd_amd
from hvd_amd import _lib as L, multigpu as M, synth
lib = L.init(0)
n = int(os.environ.get("N", 200_000))
db, _ = synth.hash_db(n, seed=3, plant_fraction=0.02)
d_db = L.DeviceBuffer.from_array(db)
d_img = M.expand_fp4(d_db.ptr, n)
cap = 1 << 20
d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
res = {}
for v in (12, int(sys.argv[1]) if len(sys.argv) > 1 else 15):
    d_cnt.zero()
    M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, cap, d_cnt.ptr, v)
    cnt = int(d_cnt.to_array(np.uint64, 1)[0])
    got = d_pairs.to_array(L.PAIR_DTYPE, cnt)
    res[v] = set((int(a), int(b), int(c)) for a, b, c in zip(got["i"], got["j"], got["dist"]))
    print(v, cnt, len(res[v]))
vs = list(res)
miss = sorted(res[vs[0]] - res[vs[1]]); extra = sorted(res[vs[1]] - res[vs[0]])
print("missing", len(miss), "extra", len(extra))
bits = np.unpackbits(db, axis=1, bitorder="little")
for i, j, d in miss[:40]:
    x = bits[i] ^ bits[j]
    print(f"i={i} j={j} d={d} lo={x[:128].sum()} hi={x[128:].sum()} i%32={i%32} j%32={j%32} i%1024={i%1024} rowreg={(i%32)%4 + 4*((i%32)//8)} h={((i%32)//4)%2}")
