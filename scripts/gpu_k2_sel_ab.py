"""Dev A/B (round 5): config-5 video search and the uniform 1M headline with the first stage's selection forced to bits 0..127 /
128..255 / 0..63+192..255 and left to the probe."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hvd_amd import _lib as L, pipeline, synth, multigpu as M
lib = L.init(0)


def get(key):
    v = C.c_int(0); L.check(lib.hvd_debug_get(key, C.byref(v))); return v.value


def timed(fn, reps=4):
    ms = []
    for r in range(reps + 1):
        L.check(lib.hvd_timer_start()); out = fn(); t = C.c_float(0); L.check(lib.hvd_timer_stop(C.byref(t)))
        if r: ms.append(t.value)
    return out, float(np.mean(ms)), float(np.std(ms))


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
_, recs0, lib5 = pipeline.dedupe_frames_on_device(d_frames.ptr, np.arange(V + 1, dtype=np.int64) * F, 64, 64, 1, keep_library=True)
d_frames.free()
print("config 5:", lib5.n_frames, "kept frames,", len(recs0), "records")
n1 = 1_000_000
db, _ = synth.hash_db(n1, seed=3)
d_db = L.DeviceBuffer.from_array(db)
d_img = M.expand_fp4(d_db.ptr, n1)
d_pairs, d_cnt = L.DeviceBuffer(16 << 20), L.DeviceBuffer(8)
for rnd in range(2):
    for sel in (-1, 0, 1, 2, -1):
        L.check(lib.hvd_debug_set(b"mfma_force_sel", sel))
        recs, ms, sd = timed(lambda: lib5.match_videos())
        assert np.array_equal(recs, recs0)
        line = f"sel {sel:2d}: cfg5 search {ms:7.2f} +- {sd:4.2f} ms form {get(b'mfma_auto_form')} ran on {get(b'mfma_auto_half')} survivors lo/hi/mix {get(b'mfma_probe_survivors')}/{get(b'mfma_probe_survivors_hi')}/{get(b'mfma_probe_survivors_mix')}"
        for v in (13, 9, 18):
            def one():
                d_cnt.zero()
                M.launch_allpairs(lib, d_db.ptr, d_img.ptr, n1, None, 31, 0, 1, d_pairs.ptr, 1 << 20, d_cnt.ptr, v)
            _, ms1, sd1 = timed(one, reps=6)
            assert int(d_cnt.to_array(np.uint64, 1)[0]) == 781
            line += f" | 1M v{v} {ms1:6.2f}"
        print(line, flush=True)
for sel in (2, 0):
    L.check(lib.hvd_debug_set(b"mfma_force_sel", sel))
    for vv in (18, 17, 19, 15, 12, 9, 18):
        L.check(lib.hvd_debug_set(b"vmatch_variant", vv))
        recs, ms, sd = timed(lambda: lib5.match_videos())
        assert np.array_equal(recs, recs0)
        print(f"sel {sel} vmatch_variant {vv}: cfg5 search {ms:7.2f} +- {sd:4.2f} ms", flush=True)
L.check(lib.hvd_debug_set(b"vmatch_variant", 0))
L.check(lib.hvd_debug_set(b"mfma_force_sel", -1))
