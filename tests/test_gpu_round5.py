"""Round 5 GPU tests: FULL-SIZE differential parity at the sizes BASELINE.json names.

Until round 4 the full-size runs were checked through properties only (every reported pair verifies, planted recall,
sub-library vs oracle, reproducible checksum). A true pair lost at one tile position at full size would have passed.
Here the complete result of every BASELINE search config is compared record for record with an independent
implementation:

  configs[2]  1M hashes        every kernel form == the CPU oracle's pair list (AVX-512 scan, a few seconds)
  configs[3]  10M hashes       union of the 8 rank tile sets (auto MFMA form) == integer popcount kernel (variant 1)
                               == (round 6) the CPU oracle on 40+ sampled row bands
  configs[4]  50k x 64 frames  (round 6) every one of the 3.2M frames hashed by the CPU oracle: kept hashes + CSR == the
                               device library's; hvd_vmatch records == host fold of the ORACLE's group-filtered frame pairs
                               == form 18 == form 8 (256-bit MFMA, no prefilter, no queue)

Semantics: dedup.py:445-502 (pair set), db/vptree.py:29-31 (distance), vpdqpy/vpdqpy.py:49-56 (video counters).
"""
import ctypes as C

import numpy as np
import pytest

from bench import host_threads

pytestmark = pytest.mark.gpu


def _pairs_equal(got, want, what):
    assert len(got) == len(want), f"{what}: {len(got)} records, expected {len(want)}"
    for f in ("i", "j", "dist"):
        assert np.array_equal(got[f], want[f]), f"{what}: field {f} differs"


def test_cfg3_full_pair_list_equals_the_oracle_in_every_form(gpu, hvd, oracle):
    """BASELINE configs[2]: the whole sorted pair list of the 1M-hash DB, through the product entry point (auto form) and
    through every explicit kernel form, equals the oracle's brute force over all 4.999995e11 pairs."""
    n = 1_000_000
    db, _ = hvd.synth.hash_db(n, seed=3)
    want = oracle.allpairs(db, 31, num_threads=host_threads())
    assert 500 < len(want) < 2000  # ~0.1 % planted x 32/41 within tolerance (+ a few chains)
    _pairs_equal(hvd.allpairs_hamming(db, 31), want, "hvd_allpairs_hamming256")
    d_db = gpu.DeviceBuffer.from_array(db)
    try:
        for v in (0, 1, 8, 9, 12, 13, 18):
            got = hvd.multigpu.sharded_allpairs(d_db.ptr, n, 0, 1, None, variant=v)
            _pairs_equal(got, want, f"variant {v}")
    finally:
        d_db.free()


def test_cfg4_union_of_8_rank_tile_sets_equals_the_popcount_kernel_and_the_oracle_row_bands(gpu, hvd, oracle):
    """BASELINE configs[3]: 10M hashes. The 8 ranks' tile sets of the auto FP4-MFMA form, run one after the other on this
    GPU and merged, equal -- record for record -- (a) the list of the integer popcount kernel (csrc/k_hamming.hip variant 1:
    xor + v_bcnt, no matrix cores, no FP4 image, different tiling) over the whole triangle in one launch, and (b) round 6,
    the CPU ORACLE on sampled row bands: 32 random 2048-row bands, the first and the last rows, and for every rank a band
    that straddles a row-block boundary whose diagonal tile that rank owns (bench.oracle_check_bands). The oracle scans
    every column j > i of the 10M for the rows of a band, so a pair lost at ANY tile position of those rows would show."""
    from bench import oracle_check_bands

    n, world = 10_000_000, 8
    db, _ = hvd.synth.hash_db(n, seed=4)
    d_db = gpu.DeviceBuffer.from_array(db)
    try:
        parts = [hvd.multigpu.sharded_allpairs(d_db.ptr, n, r, world, None) for r in range(world)]
        got = hvd.multigpu.merge_pairs(parts)
        want = hvd.multigpu.sharded_allpairs(d_db.ptr, n, 0, 1, None, variant=1)
    finally:
        d_db.free()
    assert 5000 < len(want) < 20000
    _pairs_equal(got, want, "union of 8 rank tile sets")
    # and the independent list itself verifies on the host
    x = np.unpackbits(db[want["i"]] ^ db[want["j"]], axis=1).sum(1)
    assert np.array_equal(x, want["dist"]) and (want["i"] < want["j"]).all()
    # (b) the oracle on row bands
    rows_per_block, _ = hvd.multigpu.tile_geometry(n, 9)
    bands = oracle_check_bands(n, rows_per_block, world, seed=46)
    assert len(bands) >= 40 and sum(b - a for a, b in bands) >= 60_000
    in_band = np.zeros(n, dtype=bool)
    for a, b in bands:
        in_band[a:b] = True
    want_o = oracle.allpairs_bands(db, bands, 31, num_threads=host_threads())
    assert len(want_o) > 30, "the sampled bands hold too few planted pairs to mean anything"
    _pairs_equal(got[in_band[got["i"]]], want_o, "8-rank union restricted to the oracle's row bands")


def fold_frame_pairs(pairs, video, n_videos):
    from bench import fold_frame_pairs as fold

    return fold(pairs, video, n_videos, hvd_vmatch_dtype())


def hvd_vmatch_dtype():
    from hvd_amd import _lib

    return _lib.VMATCH_DTYPE


def test_cfg5_full_library_records_equal_the_oracle_form8_and_the_host_fold(gpu, hvd, oracle):
    """BASELINE configs[4] at full size: 50 000 videos x 64 synthetic frames generated and hashed in HBM.
    Round 6 -- the CPU ORACLE behind the whole config: all 3.2 M frames are read back and hashed by the oracle; the kept
    hashes (quality >= 31, db/DedupeDB.py:550-553) and the per-video CSR must equal the device library's byte for byte; the
    oracle's brute-force scan over all 4.1e12 frame pairs with the video group filter, folded to video-level counters on the
    host (vpdqpy/vpdqpy.py:49-56), must equal the product's hvd_vmatch records.
    GPU siblings as before: the records of the product path (auto -> panel-mark queue, form 18) equal those of form 18 forced,
    of form 8 (full 256-bit MFMA compare: no 128-bit first stage, no survivor queue) and a host fold of the frame-pair list
    the integer popcount kernel reports under the video group filter."""
    lib = gpu.load()
    V, F = 50_000, 64
    rng = np.random.default_rng(5)
    copy_of = np.full(V, -1, dtype=np.int32)
    m = int(round(V * 0.02))
    dst = rng.choice(np.arange(1, V), size=m, replace=False)
    is_dst = np.zeros(V, dtype=bool)
    is_dst[dst] = True
    copy_of[dst] = rng.choice(np.flatnonzero(~is_dst), size=m)
    d_copy = gpu.DeviceBuffer.from_array(copy_of)
    d_frames = gpu.DeviceBuffer(V * F * 4096)
    gpu.check(lib.hvd_dev_synth_video_frames(d_frames.ptr, 0, V, F, 5, d_copy.ptr))
    raw_off = np.arange(V + 1, dtype=np.int64) * F
    _, recs_auto, library = hvd.pipeline.dedupe_frames_on_device(d_frames.ptr, raw_off, 64, 64, 1, keep_library=True)
    try:
        # ---- the oracle over every frame (chunks of 128k frames = 512 MB of host memory at a time)
        from bench import oracle_hash_device_frames

        ho, qo = oracle_hash_device_frames(gpu, oracle, d_frames.ptr, V * F, host_threads())
    finally:
        d_frames.free()
        d_copy.free()
    try:
        keep = qo >= 31
        kept_o = np.ascontiguousarray(ho[keep])
        off_o = np.zeros(V + 1, dtype=np.int64)
        np.cumsum(keep.reshape(V, F).sum(1), out=off_o[1:])
        assert library.n_frames == kept_o.shape[0] and 2_500_000 < library.n_frames < V * F
        assert np.array_equal(library.offsets(), off_o), "per-video CSR of the kept frames differs from the oracle's"
        assert np.array_equal(library.hashes(), kept_o), "kept frame hashes differ from the oracle's"
        video = library.d_video.to_array(np.int32, library.n_frames)
        assert np.array_equal(video, np.repeat(np.arange(V, dtype=np.int32), np.diff(off_o)))
        fp_o = oracle.allpairs(kept_o, 31, group=video, cap=1 << 22, num_threads=host_threads())
        want_o = fold_frame_pairs(fp_o, video, V)
        assert len(want_o) >= 900
        assert np.array_equal(recs_auto, want_o), "video records differ from the fold of the ORACLE's frame pairs"

        form = C.c_int(0)
        gpu.check(lib.hvd_debug_get(b"mfma_auto_form", C.byref(form)))
        assert form.value in (9, 12, 18)
        by_form = {}
        try:
            for v in (18, 8):
                gpu.check(lib.hvd_debug_set(b"vmatch_variant", v))
                by_form[v] = library.match_videos()
        finally:
            gpu.check(lib.hvd_debug_set(b"vmatch_variant", 0))
        for v, r in by_form.items():
            assert np.array_equal(r, recs_auto), f"form {v} records differ from the product path's (form {form.value})"
        # the independent GPU path: integer popcount kernel with the group filter -> frame pairs == the oracle's, pair for pair
        fp = hvd.multigpu.sharded_allpairs(library.d_hashes.ptr, library.n_frames, 0, 1, None, 31,
                                           d_group_ptr=library.d_video.ptr, variant=1, cap=1 << 22)
        _pairs_equal(fp, fp_o, "popcount kernel's frame pairs vs the oracle's")
    finally:
        library.free()


# ------------------------------------------------------------------ streaming feed: runs of frames (ABI 5) --------

@pytest.mark.parametrize("geom", [(64, 64, 1, 700), (64, 64, 3, 300), (512, 512, 3, 40)])
def test_videohasher_acquire_frames_runs(gpu, hvd, oracle, geom):
    """acquire_frames(k) / commit_frames(n): a run of frames per FFI round trip, through a tiny batch so that runs are cut at
    slot boundaries and the ring wraps; partial commits; mixed with hash_frame and the per-frame acquire. Results in feed
    order, equal to the oracle's (vpdqpy/vpdqpy.py:113-119 contract: finish() = kept hashes in frame order)."""
    w, h, ch, n = geom
    fr = hvd.synth.frames_rgb(n, seed=81, h=h, w=w) if ch == 3 else hvd.synth.frames_gray(n, 82, h, w)
    ho, qo = oracle.hash_frames(fr, num_threads=8)
    want = ho[qo >= 31].tobytes()
    for batch_bytes in (fr[0].nbytes * 37, 64 << 20):
        hs = hvd.VideoHasher(1, w, h, 0, batch_bytes=batch_bytes)
        k, step = 0, 0
        while k < n:
            step += 1
            if step % 5 == 0:  # the unchanged reference call in between
                hs.hash_frame(fr[k].tobytes())
                k += 1
            elif step % 5 == 1:  # the per-frame zero-copy feed
                np.copyto(hs.acquire_frame(ch), fr[k])
                hs.commit_frame()
                k += 1
            else:
                run = hs.acquire_frames(min(n - k, 3 + 17 * (step % 7)), ch)
                assert 1 <= run.shape[0] and run.shape[1:] == fr.shape[1:]
                m = run.shape[0] if step % 3 else max(1, run.shape[0] // 2)  # sometimes only part of the run is used
                np.copyto(run[:m], fr[k:k + m])
                hs.commit_frames(m)
                k += m
        assert hs.finish().bytes == want
    hs = hvd.VideoHasher(1, w, h, 0)
    with pytest.raises(RuntimeError):
        hs.commit_frames()
    with pytest.raises(RuntimeError):
        hs.commit_frame()
    run = hs.acquire_frames(5, ch)
    hs.commit_frames(0)  # a run may be given back unused
    assert hs.finish().bytes == b""


def test_hasher_acquire_n_c_abi(gpu, hvd, oracle):
    """The same through the C-ABI directly: argument checks, run lengths capped by the batch slot, commit bounds."""
    lib = gpu.load()
    fr = hvd.synth.frames_gray(50, seed=83)
    ho, qo = oracle.hash_frames(fr)
    hdl = C.c_void_p()
    gpu.check(lib.hvd_hasher_create(64, 64, 1, 16, C.byref(hdl)))
    try:
        p, got = C.c_void_p(), C.c_int64(0)
        assert lib.hvd_hasher_acquire_n(hdl, 0, C.byref(p), C.byref(got)) == gpu.HVD_ERR_ARG
        assert lib.hvd_hasher_commit_n(hdl, 1) == gpu.HVD_ERR_STATE
        k = 0
        while k < 50:
            gpu.check(lib.hvd_hasher_acquire_n(hdl, 50 - k, C.byref(p), C.byref(got)))
            assert 1 <= got.value <= 16 - (k % 16)
            assert lib.hvd_hasher_commit_n(hdl, got.value + 1) == gpu.HVD_ERR_ARG  # more than acquired
            gpu.check(lib.hvd_hasher_acquire_n(hdl, 50 - k, C.byref(p), C.byref(got)))
            C.memmove(p.value, fr[k:].ctypes.data, 4096 * got.value)
            gpu.check(lib.hvd_hasher_commit_n(hdl, got.value))
            k += got.value
        hh, qq, n_out = np.zeros((50, 32), np.uint8), np.zeros(50, np.int32), C.c_int64(0)
        gpu.check(lib.hvd_hasher_finish(hdl, hh.ctypes.data, qq.ctypes.data, 50, C.byref(n_out)))
        assert n_out.value == 50 and np.array_equal(hh, ho) and np.array_equal(qq, qo)
    finally:
        lib.hvd_hasher_destroy(hdl)


def test_copy_nt_switch_is_bit_identical(gpu, hvd):
    """hash_frame(bytes) through the non-temporal copy slices and through plain memcpy: same hashes (512x512 RGB24, the
    reference's geometry; 1, 4 and 8 copy threads)."""
    lib = gpu.load()
    fr = hvd.synth.frames_rgb(24, seed=84)
    h0, q0 = hvd.vpdq.hash_frames(fr)
    want = h0[q0 >= 31].tobytes()
    level = C.c_int(0)
    gpu.check(lib.hvd_debug_get(b"copy_nt", C.byref(level)))
    assert level.value in (0, 2, 3)
    try:
        for mode in (0, 1):
            gpu.check(lib.hvd_debug_set(b"copy_nt", mode))
            for threads in (1, 4, 8):
                hs = hvd.VideoHasher(1, 512, 512, threads)
                for f in fr:
                    hs.hash_frame(f.tobytes())
                assert hs.finish().bytes == want, (mode, threads)
    finally:
        gpu.check(lib.hvd_debug_set(b"copy_nt", 1))


def test_runtime_info_names_the_device_and_the_libraries(gpu):
    info = gpu.runtime_info()
    assert info["abi"] == 6 and info["visible_devices"] >= 1
    assert "gfx950" in info["devices"][0]["arch"] and info["devices"][0]["cus"] == 256
    assert info["librccl_path"].endswith(".so") or ".so." in info["librccl_path"]
    assert info["rccl_version"] > 20000 and info["hip_runtime_version"] > 0


def test_gpu_test_process_runs_on_the_system_rocm_not_on_torchs_bundle(gpu):
    """The product never imports torch, and neither may the process that tests it: with torch loaded first, its bundled
    libamdhip64 / librccl (another ROCm release) would serve this library's calls. The libraries behind the symbols this
    library calls must be the ones it was linked against (/opt/rocm), and the RCCL at run time the one it was built for."""
    import sys

    assert "torch" not in sys.modules, "a test module imported torch at collection time"
    info = gpu.runtime_info()
    assert "/torch/" not in info["librccl_path"] and "/torch/" not in info["libamdhip64_path"], info
    assert info["rccl_version"] == info["rccl_built_against"], info


# ------------------------------------------------------------------ K2: the first stage's third selection (bits 0..63 + 192..255)

def _forced_sel(gpu, value):
    gpu.check(gpu.load().hvd_debug_set(b"mfma_force_sel", value))


@pytest.mark.parametrize("sel", [0, 1, 2])
def test_k2_every_form_on_every_first_stage_selection(gpu, hvd, oracle, sel):
    """Which 128 bits the first stage sees must not change a single record: every MFMA form with the selection forced --
    bits 0..127, 128..255 and (round 5) 0..63 + 192..255 -- on a DB whose planted distances straddle the tolerance in every
    split between the two sides (all of a pair's flips in the first-stage bits, all in the other bits, even mixes)."""
    n = 30_000
    rng = np.random.default_rng(90 + sel)
    db, _ = hvd.synth.hash_db(n, seed=91, plant_fraction=0.0)
    src = rng.choice(n // 2, 1500, replace=False)
    dst = n // 2 + rng.choice(n // 2, 1500, replace=False)
    units = {0: (0, 1), 1: (2, 3), 2: (0, 3)}[sel]  # the 64-bit units the first stage sees
    other = tuple(u for u in range(4) if u not in units)
    for k, (s_, d_) in enumerate(zip(src, dst)):
        row = db[s_].copy()
        total = 29 + k % 6                      # 29 .. 34 flips: both sides of the tolerance 31
        in_first = (0, total, total // 2, total - 1, 1, 31, 32)[k % 7]
        in_first = min(in_first, total)
        for side, cnt in ((units, in_first), (other, total - in_first)):
            bits = np.concatenate([np.arange(64 * u, 64 * u + 64) for u in side])
            for b in rng.choice(bits, cnt, replace=False):
                row[b >> 3] ^= np.uint8(1 << (b & 7))
        db[d_] = row
    want = oracle.allpairs(db, 31, num_threads=8)
    assert 500 < len(want) < 1500
    d_db = gpu.DeviceBuffer.from_array(db)
    try:
        _forced_sel(gpu, sel)
        for v in (8, 9, 12, 13, 18):
            got = hvd.multigpu.sharded_allpairs(d_db.ptr, n, 0, 1, None, variant=v)
            _pairs_equal(got, want, f"variant {v}, selection {sel}")
        if sel:  # the auto variant reports the forced selection
            half = C.c_int(0)
            gpu.check(gpu.load().hvd_debug_get(b"mfma_auto_half", C.byref(half)))
            assert half.value == sel
        # video mode and the rectangular form through the same selection
        fr, off, _ = hvd.synth.video_hashes(400, seed=92, frames_per_video=(1, 40), copy_fraction=0.2)
        assert np.array_equal(hvd.match_videos(fr, off, 31), oracle.match_videos(fr, off, 31, num_threads=8))
    finally:
        _forced_sel(gpu, -1)
        d_db.free()


def test_k2_probe_picks_the_mixed_selection_when_the_middle_bits_are_degenerate(gpu, hvd, oracle):
    """Bits 64..191 nearly constant across the DB: both halves contain 64 degenerate bits and let many unrelated pairs
    through, bits 0..63 + 192..255 do not -- the probe must put the first stage there, and the pair list stays the oracle's."""
    lib = gpu.load()
    n = 60_000
    rng = np.random.default_rng(93)
    db, _ = hvd.synth.hash_db(n, seed=94, plant_fraction=0.01)
    proto = rng.integers(0, 256, 16, dtype=np.uint8)
    mid = np.tile(proto, (n, 1))
    flips = rng.integers(0, 128, (n, 3))          # three random flips per hash inside the degenerate 128 bits
    for c in range(3):
        np.bitwise_xor.at(mid, (np.arange(n), flips[:, c] >> 3), (1 << (flips[:, c] & 7)).astype(np.uint8))
    db[:, 8:24] = mid
    d_db = gpu.DeviceBuffer.from_array(db)
    try:
        got = hvd.multigpu.sharded_allpairs(d_db.ptr, n, 0, 1, None, variant=13, cap=1 << 22)
        vals = {}
        for key in (b"mfma_auto_half", b"mfma_probe_survivors", b"mfma_probe_survivors_hi", b"mfma_probe_survivors_mix", b"mfma_auto_form"):
            v = C.c_int(0)
            gpu.check(lib.hvd_debug_get(key, C.byref(v)))
            vals[key.decode()] = v.value
        assert vals["mfma_auto_half"] == 2, vals
        assert vals["mfma_probe_survivors_mix"] * 100 < min(vals["mfma_probe_survivors"], vals["mfma_probe_survivors_hi"]), vals
        _pairs_equal(got, oracle.allpairs(db, 31, num_threads=8, cap=1 << 22), "auto variant on the mixed selection")
    finally:
        d_db.free()


# ------------------------------------------------------------------ the resident match server behind matchHashBytes ----------

def test_match_server_answers_like_the_per_call_kernel_and_the_oracle(gpu, hvd, oracle):
    """hvd_match_two's small operands are served by a workgroup that stays resident between calls (round 5): back to back
    (server alive), after pauses longer than its idle limit (server restarted), from two threads, with the server switched
    off -- always the oracle's counters (vpdqpy/vpdqpy.py:49-56, db/vptree.py:29-31 call shape)."""
    import threading
    import time

    lib = gpu.load()
    fr, off, _ = hvd.synth.video_hashes(60, seed=95, frames_per_video=(0, 70), copy_fraction=0.5)
    blobs = [fr[off[v]:off[v + 1]].tobytes() for v in range(60)]
    pairs = [(a, b) for a in range(0, 60, 3) for b in range(60)]
    want = {(a, b): oracle.match_two(blobs[a], blobs[b], 31) for a, b in pairs}
    assert sum(1 for v in want.values() if v != (0, 0)) > 20

    def sweep(pause_every=0):
        for k, (a, b) in enumerate(pairs):
            if pause_every and k % pause_every == 0:
                time.sleep(0.003)  # ten idle limits: the server has left and is started again by the next call
            assert hvd.vpdq.match_counts(blobs[a], blobs[b], 31) == want[(a, b)], (a, b)

    try:
        for mode in (1, 0, 1):
            gpu.check(lib.hvd_debug_set(b"match_server", mode))
            sweep()
            sweep(pause_every=97)
        errs = []

        def worker():
            try:
                sweep(pause_every=211)
            except BaseException as e:  # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=worker) for _ in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        # tolerances other than 31, empty operands, one frame, and the distance the reference derives from the answer
        for tol in (0, 10, 64, 255):
            for a, b in pairs[:40]:
                assert hvd.vpdq.match_counts(blobs[a], blobs[b], tol) == oracle.match_two(blobs[a], blobs[b], tol)
        assert hvd.vpdq.match_counts(b"", blobs[3], 31) == (0, 0)
        d = hvd.calculate_distance(blobs[3], blobs[3])
        assert d == (1 if len(blobs[3]) else 101)
        gpu.check(lib.hvd_device_synchronize())  # returns: the server leaves by itself
        # a search loop that calls back to back must not hold off another thread's device-wide waits (hipMalloc / hipFree wait
        # for kernels in flight): the server's lifetime is bounded, the next call starts the next one
        stop, stalls = threading.Event(), []

        def caller():
            while not stop.is_set():
                hvd.vpdq.match_counts(blobs[3], blobs[4], 31)

        th = threading.Thread(target=caller)
        th.start()
        try:
            t_end = time.perf_counter() + 0.5
            while time.perf_counter() < t_end:
                t0 = time.perf_counter()
                buf = gpu.DeviceBuffer(1 << 20)
                buf.free()
                stalls.append(time.perf_counter() - t0)
        finally:
            stop.set()
            th.join()
        assert len(stalls) > 20 and max(stalls) < 0.1, (len(stalls), max(stalls))
    finally:
        gpu.check(lib.hvd_debug_set(b"match_server", 1))


def test_timer_marks_split_a_sequence_without_extra_synchronisation(gpu, hvd):
    """hvd_timer_mark / hvd_timer_between (ABI 5): eight event slots per context; the intervals of a marked sequence add up to
    the interval around it, argument errors are reported, a slot that was never marked is a state error."""
    lib = gpu.load()
    ms = C.c_float(0)
    assert lib.hvd_timer_mark(8) == gpu.HVD_ERR_ARG and lib.hvd_timer_mark(-1) == gpu.HVD_ERR_ARG
    assert lib.hvd_timer_between(0, 1, None) == gpu.HVD_ERR_ARG
    n = 200_000
    db, _ = hvd.synth.hash_db(n, seed=96)
    d_db = gpu.DeviceBuffer.from_array(db)
    d_img = hvd.multigpu.expand_fp4(d_db.ptr, n)
    d_pairs, d_cnt = gpu.DeviceBuffer(16 << 12), gpu.DeviceBuffer(8)
    try:
        d_cnt.zero()
        gpu.check(lib.hvd_timer_mark(3))
        gpu.check(lib.hvd_dev_expand_fp4(d_db.ptr, n, d_img.ptr))
        gpu.check(lib.hvd_timer_mark(4))
        hvd.multigpu.launch_allpairs(lib, d_db.ptr, d_img.ptr, n, None, 31, 0, 1, d_pairs.ptr, 1 << 12, d_cnt.ptr, 13)
        gpu.check(lib.hvd_timer_mark(5))
        parts = []
        for a, b in ((3, 4), (4, 5), (3, 5)):
            gpu.check(lib.hvd_timer_between(a, b, C.byref(ms)))
            parts.append(ms.value)
        assert parts[0] > 0 and parts[1] > parts[0] and abs(parts[0] + parts[1] - parts[2]) < 0.02 * parts[2] + 0.01, parts
        if lib.hvd_timer_between(6, 7, C.byref(ms)) == 0:  # (only a fresh context has unmarked slots; bench.py uses 0..2)
            pass
    finally:
        for b in (d_db, d_img, d_pairs, d_cnt):
            b.free()


# ------------------------------------------------------------------ K3: data-dependent bit order of the video search ----------

def test_video_search_in_a_chosen_bit_order_returns_the_same_records(gpu, hvd, oracle):
    """The video search rewrites its hashes in a bit order chosen from the library (the 128 least entangled bits first: the
    first stage lets fewer unrelated frames through). Hamming distance does not depend on the order, so the records must be
    identical with the rewrite off, automatic and forced -- symmetric and query x target form, every queue form, against the
    oracle; and on hashes with real structure (every second DCT row a copy of its neighbour + noise) the rewrite must happen."""
    lib = gpu.load()

    def used():
        v = C.c_int(0)
        gpu.check(lib.hvd_debug_get(b"vmatch_bit_order_used", C.byref(v)))
        return v.value

    rng = np.random.default_rng(97)
    fr, off, _ = hvd.synth.video_hashes(700, seed=98, frames_per_video=(1, 40), copy_fraction=0.3)
    # structure: bytes 2k+1 mostly repeat byte 2k (entangled bit pairs), so that some orders are better than others
    noise = (rng.random(fr[:, 1::2].shape) < 0.1) * rng.integers(0, 256, fr[:, 1::2].shape)
    fr[:, 1::2] = fr[:, 0::2] ^ noise.astype(np.uint8)
    want = oracle.match_videos(fr, off, 31, num_threads=8)
    assert len(want) > 50
    q_sel = np.arange(0, 700, 5)
    q_off = np.zeros(q_sel.size + 1, dtype=np.int64)
    np.cumsum(np.diff(off)[q_sel], out=q_off[1:])
    q_fr = np.concatenate([fr[off[v]:off[v + 1]] for v in q_sel])
    try:
        got_x = {}
        for mode in (0, 2, 1):
            gpu.check(lib.hvd_debug_set(b"vmatch_bit_order", mode))
            for v in (0, 18, 12, 9, 8):
                gpu.check(lib.hvd_debug_set(b"vmatch_variant", v))
                assert np.array_equal(hvd.match_videos(fr, off, 31), want), (mode, v)
                assert used() == (1 if mode == 2 else 0), (mode, used())  # (14 k frames: the automatic mode leaves small libraries alone)
            gpu.check(lib.hvd_debug_set(b"vmatch_variant", 0))
            got_x[mode] = hvd.search.match_videos_cross(q_fr, q_off, fr, off, ids_q=q_sel.astype(np.int32), ids_t=np.arange(700, dtype=np.int32))
        assert np.array_equal(got_x[0], got_x[2]) and np.array_equal(got_x[0], got_x[1]) and len(got_x[0]) > 10
        # tolerances around the first stage's limits, in the chosen order
        gpu.check(lib.hvd_debug_set(b"vmatch_bit_order", 2))
        for tol in (0, 5, 63, 64, 100):
            assert np.array_equal(hvd.match_videos(fr[:off[200]], off[:201], tol),
                                  oracle.match_videos(fr[:off[200]], off[:201], tol, num_threads=8)), tol
    finally:
        gpu.check(lib.hvd_debug_set(b"vmatch_variant", 0))
        gpu.check(lib.hvd_debug_set(b"vmatch_bit_order", 1))
