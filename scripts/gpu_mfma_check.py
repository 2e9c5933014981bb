"""Developer check: FP4-MFMA all-pairs variants vs the oracle + timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth, multigpu as M
from oracle import oracle as O
lib = L.init(0)

def run(db, variant, max_dist=31, group=None, cap=1 << 20, reps=1):
    n = len(db)
    d_db = L.DeviceBuffer.from_array(db)
    sz = C.c_size_t(0); L.check(lib.hvd_fp4_image_bytes(n, C.byref(sz)))
    d_img = L.DeviceBuffer(sz.value)
    L.check(lib.hvd_dev_expand_fp4(d_db.ptr, n, d_img.ptr))
    d_grp = L.DeviceBuffer.from_array(group) if group is not None else None
    d_pairs = L.DeviceBuffer(16 * cap); d_cnt = L.DeviceBuffer(8)
    best = 1e9
    for r in range(reps + 1):
        d_cnt.zero()
        L.check(lib.hvd_timer_start())
        L.check(lib.hvd_dev_allpairs_hamming256_mfma(d_db.ptr, d_img.ptr, n, d_grp.ptr if d_grp else None, max_dist, 0, 1,
                                                     d_pairs.ptr, cap, d_cnt.ptr, variant))
        ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
        if r or reps == 0: best = min(best, ms.value)
    cnt = int(d_cnt.to_array(np.uint64, 1)[0])
    recs = M.merge_pairs([d_pairs.to_array(L.PAIR_DTYPE, min(cnt, cap))])
    return recs, cnt, best

for n in (2, 33, 1000, 1025, 5000, 20000):
    db, _ = synth.hash_db(n, seed=70 + n % 5, plant_fraction=0.02)
    want = O.allpairs(db, 31, num_threads=8)
    for v in (8, 9, 10, 11):
        got, cnt, _ = run(db, v, reps=0)
        print(f"n={n} variant={v}: count {cnt} oracle {len(want)} equal {np.array_equal(got, want)}")
db, _ = synth.hash_db(600, seed=50, plant_fraction=0.05)
for md in (0, 1, 30, 31, 32, 63, 64, 100, 127, 128, 256):
    want = O.allpairs(db, md, cap=600 * 600)
    for v in (8, 9):
        got, cnt, _ = run(db, v, max_dist=md, reps=0)
        print(f"max_dist={md} variant={v}: equal {np.array_equal(got, want)} ({cnt})")
db, _ = synth.hash_db(5000, seed=52, plant_fraction=0.05)
grp = (np.arange(5000) // 3).astype(np.int32)
want = O.allpairs(db, 31, group=grp, num_threads=8)
print("group filter:", [np.array_equal(run(db, v, group=grp, reps=0)[0], want) for v in (8, 9, 10, 11)])
db[100:140] = db[7]
want = O.allpairs(db, 31, num_threads=8)
print("duplicates:", [np.array_equal(run(db, v, reps=0)[0], want) for v in (8, 9, 10, 11)])

for n in (200_000, 1_000_000):
    db, _ = synth.hash_db(n, seed=3)
    for v in (8, 9, 10, 11):
        got, cnt, ms = run(db, v, reps=2)
        print(f"n={n} variant={v}: {ms:.2f} ms  {n * (n - 1) / 2 / ms / 1e9:.2f} Tcmp/s  pairs={cnt}")
