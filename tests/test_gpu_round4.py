"""Round-4 GPU parity tests: the pair-queue form of the FP4-MFMA all-pairs kernel (variant 15: first-stage survivors
settled pair by pair on the VALU instead of tile by tile on the matrix pipe) -- chosen by the probe on real frame hashes,
in frame-pair, video and cross mode, with and without packed hashes, and on the data that overflows its queues -- all
through the C-ABI, against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sorted_pairs(p):
    return p[np.lexsort((p["j"], p["i"]))]


def _auto(gpu, key):
    v = C.c_int(0)
    gpu.check(gpu.load().hvd_debug_get(key, C.byref(v)))
    return v.value


def _run(gpu, hvd, db, variant, group=None, max_dist=31, cap=1 << 16):
    n = len(db)
    d_db = gpu.DeviceBuffer.from_array(db)
    d_img = hvd.multigpu.expand_fp4(d_db.ptr, n)
    d_grp = gpu.DeviceBuffer.from_array(group) if group is not None else None
    d_pairs, d_cnt = gpu.DeviceBuffer(16 * cap), gpu.DeviceBuffer(8)
    d_cnt.zero()
    hvd.multigpu.launch_allpairs(gpu.load(), d_db.ptr, d_img.ptr, n, d_grp.ptr if d_grp else None, max_dist, 0, 1,
                                 d_pairs.ptr, cap, d_cnt.ptr, variant)
    cnt = int(d_cnt.to_array(np.uint64, 1)[0])
    assert cnt <= cap
    got = _sorted_pairs(d_pairs.to_array(gpu.PAIR_DTYPE, cnt))
    for b in (d_db, d_img, d_pairs, d_cnt, d_grp):
        if b is not None:
            b.free()
    return got


@pytest.fixture(scope="module")
def frame_library(gpu, hvd):
    """Hashes of synthetic VIDEO FRAMES (the config-5 generator): the data on which the first 128 bits of unrelated hashes
    agree within the tolerance for ~2e-4 of all pairs. 1 500 videos x 64 frames, 2 % planted copies."""
    from hvd_amd import pipeline
    lib = gpu.load()
    V, F = 1500, 64
    rng = np.random.default_rng(41)
    copy_of = np.full(V, -1, dtype=np.int32)
    dst = rng.choice(np.arange(V // 2, V), V // 50, replace=False)
    copy_of[dst] = rng.integers(0, V // 2, dst.size)
    d_copy = gpu.DeviceBuffer.from_array(copy_of)
    d_frames = gpu.DeviceBuffer(V * F * 4096)
    gpu.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, V, F, 5, d_copy.ptr))
    d_h, d_q = pipeline.hash_frames_on_device(d_frames.ptr, V * F, 64, 64, 1)
    libr = pipeline.DeviceLibrary.from_raw_hashes(d_h.ptr, d_q.ptr, V * F, np.arange(V + 1, dtype=np.int64) * F)
    for b in (d_frames, d_copy, d_h, d_q):
        b.free()
    frames, offsets = libr.hashes(), libr.offsets()
    video = np.repeat(np.arange(V, dtype=np.int32), np.diff(offsets))
    yield frames, offsets, video, libr
    libr.free()


def test_k2_probe_picks_the_pair_queue_on_frame_hashes(gpu, hvd, oracle, frame_library):
    frames, offsets, video, _ = frame_library
    want = oracle.allpairs(frames, 31, group=video, num_threads=8, cap=1 << 22)
    assert len(want) > 1000  # the planted copies
    got = _run(gpu, hvd, frames, 13, group=video, cap=len(want) + 16)
    assert _auto(gpu, b"mfma_auto_form") == 18 and _auto(gpu, b"mfma_probe_survivors") > 100
    assert np.array_equal(got, want)
    for v in (18, 12, 9):
        assert np.array_equal(_run(gpu, hvd, frames, v, group=video, cap=len(want) + 16), want), v
    want_all = oracle.allpairs(frames, 31, num_threads=8, cap=1 << 22)
    assert np.array_equal(_run(gpu, hvd, frames, 18, cap=1 << 20), want_all)


def test_k2_probe_counts_what_a_host_restatement_of_its_sample_counts(gpu, hvd, frame_library):
    """k_prefilter_probe: up to 4096 sample rows x 4096 sample columns (strided, columns half a stride off the rows), survivors
    of a 128-bit first stage over bits 0..127 and over bits 128..255 -- the two numbers the form choice rests on."""
    frames, offsets, video, _ = frame_library
    for db in (frames, frames[:3001], np.random.default_rng(5).integers(0, 256, (5000, 32), dtype=np.uint8)):
        n = len(db)
        _run(gpu, hvd, db, 13, cap=1 << 20)
        rows = min(n, 4096)
        rstride = cstride = n // rows
        ri = np.arange(rows) * rstride
        ci = np.minimum(np.arange(rows) * cstride + cstride // 2, n - 1)
        lo = np.unpackbits(db[ri][:, None, :16] ^ db[ci][None, :, :16], axis=2).sum(2) if rows <= 1024 else None
        want = [0, 0]
        for h, sl in enumerate((slice(0, 16), slice(16, 32))):
            a, b = db[ri][:, sl], db[ci][:, sl]
            for r0 in range(0, rows, 256):  # (blocked: 4096 x 4096 x 128 bits at once would be 2 GB)
                d = np.unpackbits(a[r0:r0 + 256, None, :] ^ b[None, :, :], axis=2).sum(2)
                want[h] += int((d <= 31).sum())
        assert lo is None or int((lo <= 31).sum()) == want[0]
        assert [_auto(gpu, b"mfma_probe_survivors"), _auto(gpu, b"mfma_probe_survivors_hi")] == want, n


def test_k2_pair_queue_settles_from_the_images_when_there_are_no_packed_hashes(gpu, hvd, oracle, frame_library):
    frames, offsets, video, _ = frame_library
    sub = frames[:30000]
    want = oracle.allpairs(sub, 31, group=video[:30000], num_threads=8, cap=1 << 22)
    lib = gpu.load()
    gpu.check(lib.hvd_debug_set(b"mfma_queue_packed", 0))
    try:
        for v in (18,):
            assert np.array_equal(_run(gpu, hvd, sub, v, group=video[:30000], cap=len(want) + 16), want), v
    finally:
        gpu.check(lib.hvd_debug_set(b"mfma_queue_packed", 1))


def test_k3_video_search_runs_through_the_pair_queue(gpu, hvd, oracle, frame_library):
    frames, offsets, video, libr = frame_library
    got = libr.match_videos(31)
    assert _auto(gpu, b"mfma_auto_form") == 18
    # oracle on a sub-library (every video pair of the first 400 videos + all planted copies are checked by recall below)
    sub_v = 400
    sub = oracle.match_videos(frames[: offsets[sub_v]], offsets[: sub_v + 1], 31)
    mine = got[(got["a"] < sub_v) & (got["b"] < sub_v)]
    assert np.array_equal(mine, sub)
    # the forms agree on the whole library
    lib = gpu.load()
    gpu.check(lib.hvd_debug_set(b"mfma_auto_mid", 0))
    try:
        ref = libr.match_videos(31)
        assert _auto(gpu, b"mfma_auto_form") == 12
    finally:
        gpu.check(lib.hvd_debug_set(b"mfma_auto_mid", 18))
    assert lib.hvd_debug_set(b"mfma_auto_mid", 15) == gpu.HVD_ERR_ARG  # (round 4's first queue form: pruned in round 6)
    assert np.array_equal(got, ref) and len(got) > 20
    gpu.check(lib.hvd_debug_set(b"mfma_queue_packed", 0))
    try:
        assert np.array_equal(libr.match_videos(31), ref)
    finally:
        gpu.check(lib.hvd_debug_set(b"mfma_queue_packed", 1))


def test_k3_cross_search_runs_through_the_pair_queue(gpu, hvd, oracle, frame_library):
    frames, offsets, video, _ = frame_library
    V = len(offsets) - 1
    q_sel = np.arange(0, 300, 2)
    lengths = np.diff(offsets)
    q_off = np.zeros(q_sel.size + 1, dtype=np.int64)
    np.cumsum(lengths[q_sel], out=q_off[1:])
    q_frames = np.concatenate([frames[offsets[v]:offsets[v + 1]] for v in q_sel])
    t_v = 600
    got = hvd.search.match_videos_cross(q_frames, q_off, frames[: offsets[t_v]], offsets[: t_v + 1],
                                        ids_q=q_sel.astype(np.int32), ids_t=np.arange(t_v, dtype=np.int32))
    assert _auto(gpu, b"mfma_auto_form") == 18
    want = []
    for qi, v in enumerate(q_sel):
        a = frames[offsets[v]:offsets[v + 1]].tobytes()
        for t in range(t_v):
            if t == v:
                continue
            q, th = oracle.match_two(a, frames[offsets[t]:offsets[t + 1]].tobytes(), 31)
            if q or th:
                want.append((qi, t, q, th))
    assert got.tolist() == want


def test_k2_pair_queue_overflowing_tiles_take_the_tile_route(gpu, hvd, oracle):
    """Tiles in which many lanes hold a survivor (a cluster of near-identical hashes, the diagonal) leave the queue alone;
    queues that fill up inside one super-panel are settled before they overflow: half the DB agrees in its first 128 bits
    (every tile of that half survives the first stage in every lane), the other half is uniform."""
    rng = np.random.default_rng(77)
    n = 6000
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    db[: n // 2, :16] = rng.integers(0, 256, 16, dtype=np.uint8)  # one common lower half
    # sparse survivors as well: 3 survivors per 32x32 tile on average in the uniform half (prototypes in the lower half)
    proto = rng.integers(0, 256, (340, 16), dtype=np.uint8)
    db[n // 2:, :16] = proto[rng.integers(0, 340, n - n // 2)]
    db[100] = db[5000]
    db[4000, :] = db[3500, :]
    db[4000, 20] ^= 0x3
    want = oracle.allpairs(db, 31, num_threads=8, cap=1 << 22)
    assert len(want) >= 2
    for v in (18, 12):
        assert np.array_equal(_run(gpu, hvd, db, v, cap=len(want) + 16), want), v


# ---------------------------------------------------------------- streaming hasher ----

def test_videohasher_two_threads_with_different_copy_thread_counts(gpu, hvd, oracle):
    """Two threads push 512x512 RGB24 frames through hashers with num_threads 2 and 8 (the copy pool is one per process:
    the pushers alternate job by job with different numbers of participants -- the setting of round 3's generation race),
    one of them through the reference's geometry in gray as well; every hash of every video is compared."""
    import threading

    rgb = hvd.synth.frames_rgb(8, seed=52)
    want_h, want_q = oracle.hash_frames(rgb)
    assert (want_q >= 31).sum() >= 4  # (finish() drops the others, vpdqpy/vpdqpy.py:119 / db/DedupeDB.py:550-553)
    errors = []

    def pusher(num_threads, videos, frames_per_video, phase):
        try:
            blobs = [f.tobytes() for f in rgb]
            for v in range(videos):
                h = hvd.VideoHasher(1, 512, 512, num_threads)
                order = [(v * 3 + phase + k) % 8 for k in range(frames_per_video)]
                for k in order:
                    h.hash_frame(blobs[k])
                order = np.array(order)
                if h.finish().bytes != want_h[order][want_q[order] >= 31].tobytes():
                    errors.append((num_threads, v))
                    return
        except Exception as exc:  # noqa: BLE001 - reported below
            errors.append((num_threads, repr(exc)))

    ts = [threading.Thread(target=pusher, args=(2, 12, 150, 0)), threading.Thread(target=pusher, args=(8, 12, 150, 5))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_videohasher_two_hours_at_one_frame_per_second(gpu, hvd, oracle):
    """7 200 frames through ONE VideoHasher(1, 512, 512, 0) -- the longest video the reference's defaults produce in two
    hours (vpdqpy/vpdqpy.py:72-77: one frame per second); the ring had been tested with 1 000 frames at most."""
    rgb = hvd.synth.frames_rgb(16, seed=53)
    want_h, want_q = oracle.hash_frames(rgb)
    n = 7200
    order = (np.arange(n) * 7 + (np.arange(n) // 16)) % 16
    h = hvd.VideoHasher(1, 512, 512, 0)
    blobs = [f.tobytes() for f in rgb]
    for k in order:
        h.hash_frame(blobs[k])
    got = h.finish()
    keep = want_q[order] >= 31
    assert len(got) == int(keep.sum()) and got.bytes == want_h[order][keep].tobytes()


def test_reference_boundary_vectors_replay_on_the_gpu(gpu, hvd):
    """tests/golden/import_reference.py --write (a container that has the hvdaccelerators wheel) stores the reference's own
    matchHash answers for frame pairs at exactly 30 / 31 / 32 bits and for multi-frame pairs that hinge on a 31-bit pair.
    Where that fixture exists it is replayed here through the GPU path with the reference's comparator and reduction; where
    it does not (this repository as committed: the wheel is not installable offline) parity stays UNPINNED and the test says
    so by skipping."""
    import os

    from conftest import GOLDEN

    path = os.path.join(GOLDEN, "reference_boundary.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/reference_boundary.npz absent: run tests/golden/import_reference.py --write where the wheel exists")
    g = np.load(path)
    tol = int(g["tolerance"][0])
    for x, y, want in zip(g["one_q"], g["one_t"], g["one_similarity"]):
        got = hvd.matchHashBytes(x.tobytes(), y.tobytes(), tol)
        assert abs(float(got) - float(want)) < 1e-6, (float(got), float(want))
    for qv, tv, want in zip(g["multi_q"], g["multi_t"], g["multi_similarity"]):
        got = hvd.matchHash(hvd.VpdqHash(qv.tobytes()), hvd.VpdqHash(tv.tobytes()), tol)
        assert abs(float(got) - float(want)) < 1e-6, (float(got), float(want))


def test_k1_next_frame_prefetch_switch_is_bit_identical(gpu, hvd, oracle):
    """The static launches of the 64x64 hash kernel can fetch the next frame one frame ahead (hvd_debug_set
    "pdq_hash_prefetch"; off by default: measured slower, profiles/r04_k1_grid.txt). Either way the bits are the oracle's,
    at a size where workgroups make several trips and at one where the last trip is ragged."""
    lib = gpu.load()
    for n in (10_000, 4_099):
        fr = hvd.synth.frames_gray(n, seed=71)
        wh, wq = oracle.hash_frames(fr, num_threads=8)
        for pref in (1, 0):
            gpu.check(lib.hvd_debug_set(b"pdq_hash_prefetch", pref))
            try:
                h, q = hvd.vpdq.hash_frames(fr)
            finally:
                gpu.check(lib.hvd_debug_set(b"pdq_hash_prefetch", 0))
            assert np.array_equal(h, wh) and np.array_equal(q, wq), (n, pref)


# ---------------------------------------------------------------- randomised sweeps (the dev tools, a few seeds each) ----

@pytest.mark.gpu
@pytest.mark.parametrize("script,args", [("gpu_fuzz_k2.py", ["20", "12000"]), ("gpu_fuzz_k3.py", ["10", "1000"])])
def test_randomised_parity_sweeps_of_the_pair_and_video_searches(script, args):
    """scripts/gpu_fuzz_k2.py (frame pairs, every kernel form drawn at random) and scripts/gpu_fuzz_k3.py (video-level and
    cross search through every form: ragged libraries with empty, one-frame and > 1024-frame videos, prototype-built halves,
    copies around the tolerance) against the oracle; the long runs are recorded under profiles/."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, "scripts", script)] + args, capture_output=True, text=True, timeout=1200)
    assert run.returncode == 0 and " 0 mismatches" in run.stdout, (run.stdout[-1500:], run.stderr[-1500:])
