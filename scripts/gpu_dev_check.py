"""Developer check run on the GPU box: kernels vs oracle + quick timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import hvd_amd
from hvd_amd import _lib as L, synth
from oracle import oracle as O

lib = L.init(0)
print("devices:", L.device_count())

# --- DCT matrix
d = np.zeros((16, 64), np.float32); L.check(lib.hvd_dct_matrix(d.ctypes.data))
print("dct matches oracle:", np.array_equal(d.view(np.uint32), O.dct_matrix().view(np.uint32)))

# --- K1 gray 64x64
fr = synth.frames_gray(2000, seed=2)
H = np.zeros((len(fr), 32), np.uint8); Q = np.zeros(len(fr), np.int32)
L.check(lib.hvd_pdq_hash_frames_gray_u8(fr.ctypes.data, len(fr), 64, 64, H.ctypes.data, Q.ctypes.data))
Ho, Qo = O.hash_frames(fr, num_threads=8)
print("K1 gray64: hash mismatches", int((H != Ho).any(1).sum()), "quality mismatches", int((Q != Qo).sum()),
      "quality hist", np.histogram(Qo, bins=[0, 1, 31, 50, 100, 101])[0])

# --- K1 rgb 512
fr2 = synth.frames_rgb(6, seed=6)
H2 = np.zeros((len(fr2), 32), np.uint8); Q2 = np.zeros(len(fr2), np.int32)
L.check(lib.hvd_pdq_hash_frames_rgb24_u8(fr2.ctypes.data, len(fr2), 512, 512, H2.ctypes.data, Q2.ctypes.data))
Ho2, Qo2 = O.hash_frames(fr2, num_threads=8)
print("K1 rgb512: hash mismatches", int((H2 != Ho2).any(1).sum()), "quality mismatches", int((Q2 != Qo2).sum()), Qo2)
fr3 = synth.frames_rgb(3, seed=7, h=100, w=333)
H3 = np.zeros((len(fr3), 32), np.uint8); Q3 = np.zeros(len(fr3), np.int32)
L.check(lib.hvd_pdq_hash_frames_rgb24_u8(fr3.ctypes.data, len(fr3), 100, 333, H3.ctypes.data, Q3.ctypes.data))
Ho3, Qo3 = O.hash_frames(fr3)
print("K1 rgb 100x333: hash mismatches", int((H3 != Ho3).any(1).sum()), "quality mismatches", int((Q3 != Qo3).sum()))

# --- K2 parity at 20k
db, planted = synth.hash_db(20000, seed=3, plant_fraction=0.01)
out = np.zeros(100000, L.PAIR_DTYPE); cnt = C.c_int64(0)
L.check(lib.hvd_allpairs_hamming256(db.ctypes.data, len(db), None, 31, out.ctypes.data, len(out), C.byref(cnt)))
ref = O.allpairs(db, 31, num_threads=8)
got = out[:cnt.value]
print("K2 20k: count", cnt.value, "oracle", len(ref), "equal:", np.array_equal(got, ref))

# --- K2 timing at N, variants
def time_k2(n, variant, reps=2):
    db, _ = synth.hash_db(n, seed=3)
    d_db = L.DeviceBuffer.from_array(db)
    d_pairs = L.DeviceBuffer(16 * (1 << 20)); d_cnt = L.DeviceBuffer(8)
    best = 1e9
    for r in range(reps + 1):
        d_cnt.zero()
        L.check(lib.hvd_timer_start())
        L.check(lib.hvd_dev_allpairs_hamming256(d_db.ptr, n, None, 31, 0, 1, d_pairs.ptr, 1 << 20, d_cnt.ptr, variant))
        ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
        if r > 0: best = min(best, ms.value)
    c = d_cnt.to_array(np.uint64, 1)[0]
    cmp_ = n * (n - 1) / 2
    print(f"K2 n={n} variant={variant}: {best:.2f} ms  {cmp_ / best / 1e6:.1f} Gcmp/s  pairs={c}")
    d_db.free(); d_pairs.free(); d_cnt.free()

for v in range(7):
    time_k2(200_000, v)
for v in (0, 1, 2, 3):
    time_k2(1_000_000, v, reps=1)

# --- K1 timing
def time_k1(n):
    fr = synth.frames_gray(min(n, 10000), seed=2)
    reps = (n + len(fr) - 1) // len(fr)
    fr = np.concatenate([fr] * reps)[:n]
    d_f = L.DeviceBuffer.from_array(fr); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
    best = 1e9
    for r in range(4):
        L.check(lib.hvd_timer_start())
        L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 64, 64, 1, None, d_h.ptr, d_q.ptr))
        ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
        if r > 0: best = min(best, ms.value)
    print(f"K1 n={n}: {best:.3f} ms  {n / best / 1e3:.2f} Mframes/s  ({n * 4132 / best / 1e6:.1f} GB/s algorithmic)")
for n in (10_000, 100_000, 400_000):
    time_k1(n)

# rgb512 timing
n = 256
fr = synth.frames_rgb(8, seed=6); fr = np.concatenate([fr] * (n // 8))
sb = C.c_size_t(0); L.check(lib.hvd_pdq_scratch_bytes(n, 512, 512, 3, C.byref(sb)))
d_f = L.DeviceBuffer.from_array(fr); d_s = L.DeviceBuffer(sb.value); d_h = L.DeviceBuffer(32 * n); d_q = L.DeviceBuffer(4 * n)
for r in range(3):
    L.check(lib.hvd_timer_start())
    L.check(lib.hvd_dev_pdq_hash_frames(d_f.ptr, n, 512, 512, 3, d_s.ptr, d_h.ptr, d_q.ptr))
    ms = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(ms)))
print(f"K1 rgb512 n={n}: {ms.value:.3f} ms  {n / ms.value:.1f} kframes/s  ({n * 786468 / ms.value / 1e6:.1f} GB/s algorithmic)")
